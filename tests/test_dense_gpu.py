"""GPU parity of the fp32-MFMA kernels against the torch-CPU oracle (oracle/dense_ref.py).

Tolerance: fp32 features within 1e-4 (BASELINE.json north_star), written as
atol=1e-4, rtol=1e-4 on O(1) activations.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dense_ref as R
from oracle import index_ops as O
from ptt_amd import ops, synth
from tests.util import fold_layers, mlp_layers, transformer_params

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("rows,K,Cout,relu,res", [(77, 256, 512, False, False), (256, 512, 1536, False, False),
                                                  (130, 512, 256, False, True), (64, 3, 64, True, False),
                                                  (33, 131, 96, True, True)])
def test_linear(dev, rows, K, Cout, relu, res):
    rs = np.random.RandomState(rows + K)
    x = torch.from_numpy(rs.standard_normal((rows, K)).astype(np.float32))
    w = torch.from_numpy((rs.standard_normal((Cout, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(Cout).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((rows, Cout)).astype(np.float32)) if res else None
    ref = F.linear(x, w, b)
    if relu:
        ref = F.relu(ref)
    if res:
        ref = ref + r
    wp = ops.pack_weight(w.to(dev))
    got = ops.linear(x.to(dev), wp, Cout, None, b.to(dev), relu, r.to(dev) if res else None)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("rows,widths,res,folded", [
    (77, [259, 256, 256, 259], True, True),        # vote_layer: (xyz | features) -> offsets + feature residual, 9 column tiles
    (6144, [256, 256, 256, 1], False, True),       # cla_layer at 48 frames
    (130, [256, 256, 256, 5], False, False),       # refine_layer, BatchNorm scale kept separate
    (128, [256, 256, 256], False, True),           # CosineSimAug's trailing convolutions at one frame
    (33, [40, 64], False, False), (500, [12, 96, 33, 200, 384], True, True)])
def test_rows_mlp_one_launch_equals_the_layer_chain(dev, rows, widths, res, folded):
    """ptt_rows_mlp_f32 (a Conv1d stack in one launch) against float64 torch on the CPU: every layer linear + per-channel
    scale / shift (a folded BatchNorm) + ReLU except the last, odd widths, rows that do not fill the last tile."""
    rs = np.random.RandomState(rows + len(widths))
    x = torch.from_numpy(rs.standard_normal((rows, widths[0])).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((rows, widths[-1])).astype(np.float32)) if res else None
    ref = x.double()
    layers = []
    for i, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
        w = torch.from_numpy((rs.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32))
        sc = torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
        sh = torch.from_numpy(rs.standard_normal(cout).astype(np.float32))
        relu = i < len(widths) - 2
        ref = ref @ w.double().t() * sc.double() + sh.double()
        if relu:
            ref = ref.clamp_min(0)
        if folded:
            layers.append((ops.pack_weight((w * sc[:, None]).to(dev)), None, sh.to(dev), cin, cout, relu))
        else:
            layers.append((ops.pack_weight(w.to(dev)), sc.to(dev), sh.to(dev), cin, cout, relu))
    if res:
        ref = ref + r.double()
    got = ops.rows_mlp(x.to(dev), layers, r.to(dev) if res else None)
    assert tuple(got.shape) == (rows, widths[-1])
    np.testing.assert_allclose(got.cpu().numpy(), ref.float().numpy(), **TOL)


def test_pack_weight_is_transpose_detecting(dev):
    """A = I against an asymmetric W: the GEMM must return W^T rows exactly."""
    K, Cout = 40, 64
    w = torch.arange(Cout * K, dtype=torch.float32).reshape(Cout, K) / 7.0
    x = torch.eye(K)
    got = ops.linear(x.to(dev), ops.pack_weight(w.to(dev)), Cout)
    np.testing.assert_array_equal(got.cpu().numpy(), w.t().numpy())


SA_CASES = [
    # N, M, C, spec, radius, ns        (the four SA shapes of tools/cfgs/kitti_models/ptt.yaml)
    (1024, 512, 0, [3, 64, 64, 128], 0.3, 32),
    (512, 256, 128, [131, 128, 128, 256], 0.5, 32),
    (256, 128, 256, [259, 128, 128, 256], 0.7, 32),
    (128, 64, 257, [260, 256, 256, 256], 0.3, 16),
    (200, 50, 5, [8, 32, 96], 0.4, 16),
    (300, 40, 12, [15, 64, 128], 0.6, 64),      # nsample 64: one centre spans both row tiles
]


@pytest.mark.parametrize("N,M,C,spec,radius,ns", SA_CASES)
@pytest.mark.parametrize("point_major,scale_in_weights", [(False, False), (True, False), (True, True)])
def test_sa_fused(dev, N, M, C, spec, radius, ns, point_major, scale_in_weights):
    B = 3
    rs = np.random.RandomState(N + C)
    s, _ = synth.frames(N, B, N, 64, K_s=max(16, N // 2))
    s[2] = 0.0
    xyz = torch.from_numpy(s)
    inds = torch.from_numpy(O.fps(s, M))
    new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    feats = torch.from_numpy(rs.standard_normal((B, C, N)).astype(np.float32)) if C else None
    layers = mlp_layers(N, spec)
    grouped, _, idx = R.query_and_group(xyz, new_xyz, feats, radius, ns, True, True)
    ref = F.max_pool2d(R.shared_mlp_eval(grouped, layers), kernel_size=[1, ns]).squeeze(-1)

    f_dev = None
    if feats is not None:
        f_dev = feats.to(dev)
        if point_major:   # (B,C,N) view of (B,N,C) storage: the coalesced gather path
            f_dev = f_dev.transpose(1, 2).contiguous().transpose(1, 2)
    idx_dev = ops.ball_query(new_xyz.to(dev), xyz.to(dev), radius, ns)
    np.testing.assert_array_equal(idx_dev.cpu().numpy(), idx.numpy())
    got = ops.sa_fused_forward(xyz.to(dev), new_xyz.to(dev), idx_dev, f_dev,
                               fold_layers(layers, dev, ops, scale_in_weights), radius, True, True,
                               point_major_out=point_major)
    assert tuple(got.shape) == (B, spec[-1], M)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("B,scale_in_weights", [(3, False), (3, True), (26, True)])
@pytest.mark.parametrize("N,M,C,spec,radius,ns", [c for c in SA_CASES if c[2] > 0])
def test_sa_fused_hoisted_layer0(dev, N, M, C, spec, radius, ns, B, scale_in_weights):
    """ptt_sa_desc.l0_*: layer 0's feature half evaluated once per point on the linear kernel, the kernel adds the
    three relative-coordinate terms — same oracle, same tolerance as the in-kernel layer 0. With the BatchNorm scale
    folded into the weights (what the modules pass) the 128 -> 128 -> 256 levels run on the persistent
    sa_stream_kernel; B = 26 gives its workgroups several tiles each (and a ragged last chunk), B = 3 one."""
    rs = np.random.RandomState(N + C)
    s, _ = synth.frames(N, B, N, 64, K_s=max(16, N // 2))
    s[2] = 0.0
    xyz = torch.from_numpy(s)
    inds = torch.from_numpy(O.fps(s, M))
    new_xyz = torch.gather(xyz, 1, inds.long()[..., None].expand(-1, -1, 3)).contiguous()
    feats = torch.from_numpy(rs.standard_normal((B, C, N)).astype(np.float32))
    layers = mlp_layers(N, spec)
    grouped, _, idx = R.query_and_group(xyz, new_xyz, feats, radius, ns, True, True)
    ref = F.max_pool2d(R.shared_mlp_eval(grouped, layers), kernel_size=[1, ns]).squeeze(-1)

    folded = fold_layers(layers, dev, ops)
    w0 = layers[0]["conv_weight"].reshape(spec[1], spec[0]).to(dev)
    scale0, shift0 = folded[0][1], folded[0][2]
    if scale_in_weights:
        folded = fold_layers(layers, dev, ops, scale_in_weights=True)
    rows = feats.to(dev).transpose(1, 2).contiguous()                                   # (B,N,C)
    term = ops.linear(rows, ops.pack_weight(w0[:, 3:].contiguous()), spec[1], scale0, shift0, relu=False)
    wx = (w0[:, 0:3] * scale0[:, None]).t().contiguous()                                # (3,C0)
    idx_dev = ops.ball_query(new_xyz.to(dev), xyz.to(dev), radius, ns)
    got = ops.sa_fused_forward(xyz.to(dev), new_xyz.to(dev), idx_dev, None, folded[1:], radius, True, True,
                               l0=(term, wx, True))
    assert tuple(got.shape) == (B, spec[-1], M)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("B", [2, 10])      # 128 / 64 workgroups (fewer than CUs) and 640 / 320 (several per CU, staggered slots)
@pytest.mark.parametrize("N", [128, 64])
def test_transformer_pair_kernel(dev, N, B):
    D, k = 512, 16
    rs = np.random.RandomState(N)
    P = transformer_params(N)
    s, _ = synth.frames(N, B, N, 64, K_s=N)
    xyz = torch.from_numpy(s)
    feats = torch.from_numpy(rs.standard_normal((B, N, 256)).astype(np.float32))
    ref_res, ref_attn = R.transformer_block(xyz, feats, P, k)

    d = lambda n: P[n].to(dev).contiguous()
    knn = ops.knn(xyz.to(dev), k)
    x = ops.linear(feats.to(dev), ops.pack_weight(d("fc1.weight")), D, None, d("fc1.bias"))
    wqkv = torch.cat([P["w_qs.weight"], P["w_ks.weight"], P["w_vs.weight"]], 0).to(dev)
    qkv = ops.linear(x, ops.pack_weight(wqkv), 3 * D)
    res, attn = ops.pt_attn_pair(xyz.to(dev), knn, qkv, ops.pack_delta0(d("fc_delta.0.weight"), d("fc_delta.0.bias")),
                                 ops.pack_weight(d("fc_delta.2.weight")), d("fc_delta.2.bias"),
                                 ops.pack_weight(d("fc_gamma.0.weight")), d("fc_gamma.0.bias"),
                                 ops.pack_weight(d("fc_gamma.2.weight")), d("fc_gamma.2.bias"), D, True)
    out = ops.linear(res, ops.pack_weight(d("fc2.weight")), 256, None, d("fc2.bias"), False, feats.to(dev))
    np.testing.assert_allclose(attn.cpu().numpy(), ref_attn.numpy(), **TOL)
    np.testing.assert_allclose(out.cpu().numpy(), ref_res.numpy(), **TOL)
    # attn=None path gives the same res
    res2, none = ops.pt_attn_pair(xyz.to(dev), knn, qkv, ops.pack_delta0(d("fc_delta.0.weight"), d("fc_delta.0.bias")),
                                  ops.pack_weight(d("fc_delta.2.weight")), d("fc_delta.2.bias"),
                                  ops.pack_weight(d("fc_gamma.0.weight")), d("fc_gamma.0.bias"),
                                  ops.pack_weight(d("fc_gamma.2.weight")), d("fc_gamma.2.bias"), D, False)
    assert none is None
    np.testing.assert_array_equal(res2.cpu().numpy(), res.cpu().numpy())


@pytest.mark.parametrize("B,C,Ns,Nt,point_major", [(3, 256, 128, 64, True), (2, 256, 37, 70, False), (2, 300, 5, 130, True),
                                                   (1, 20, 9, 3, False)])
def test_cosine_map_equals_torch_cosine_similarity(dev, B, C, Ns, Nt, point_major):
    """ptt_cosine_map_f32 = F.cosine_similarity of every (search, template) feature pair (p2b_xcoor.py:35-36), for both
    memory layouts the features arrive in, odd search / template counts and channel counts; an all-zero feature row (norm
    clamped at eps) gives 0."""
    rs = np.random.RandomState(C + Ns)
    sf = torch.from_numpy(rs.standard_normal((B, C, Ns)).astype(np.float32))
    tf = torch.from_numpy(rs.standard_normal((B, C, Nt)).astype(np.float32))
    tf[0, :, 0] = 0.0
    ref = F.cosine_similarity(sf[:, :, :, None], tf[:, :, None, :], dim=1, eps=1e-8)          # (B,Ns,Nt)
    if point_major:
        sd, td = sf.transpose(1, 2).contiguous().to(dev).transpose(1, 2), tf.transpose(1, 2).contiguous().to(dev).transpose(1, 2)
    else:
        sd, td = sf.to(dev), tf.to(dev)
    got = ops.cosine_map(sd, td, eps=1e-8)
    assert tuple(got.shape) == (B, Ns, Nt)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=2e-6, rtol=1e-5)
    assert float(got[0, :, 0].abs().max()) == 0.0


@pytest.mark.parametrize("n1,n2", [(128, 96), (192, 64), (512, 40)])
def test_cosine_sim_aug_fused_for_other_template_sizes(dev, n1, n2):
    """The fused CosineSimAug kernel walks the template seeds in chunks of 64 (BASELINE configs[4] leaves 512 of them):
    module output against the oracle restatement of the reference op sequence, and the fused path must be the one that
    ran (no 'unfused' note)."""
    from ptt_amd.hot_path import AttrDict
    from ptt_amd.models.similarity_modules import CosineSimAug
    from tests.util import cosine_sim_params, load_cosine_sim
    mlp, conv = cosine_sim_params(77 + n1)
    m = CosineSimAug(AttrDict.wrap(dict(DEBUG=False, MLP=dict(CHANNELS=[260, 256, 256, 256], BN=True),
                                        CONV=dict(CHANNELS=[256, 256, 256], BN=True)))).eval()
    load_cosine_sim(m, mlp, conv)
    rs = np.random.RandomState(n1 + n2)
    sf = torch.from_numpy(rs.standard_normal((2, 256, n2)).astype(np.float32))
    tf = torch.from_numpy(rs.standard_normal((2, 256, n1)).astype(np.float32))
    txyz = torch.from_numpy(rs.uniform(-2, 2, (2, n1, 3)).astype(np.float32))
    ref, _ = R.cosine_sim_aug(sf, tf, txyz, mlp, conv)
    before = dict(ops.unfused_calls)
    with torch.no_grad():
        got = m.to(dev)({'search_feats': sf.to(dev), 'template_feats': tf.to(dev), 'template_seeds': txyz.to(dev)})['cosine_feats']
    assert dict(ops.unfused_calls) == before
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), **TOL)
