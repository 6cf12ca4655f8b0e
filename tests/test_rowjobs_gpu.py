"""ptt_row_jobs_f32 (ptt_amd/csrc/rowjobs.hip) through the C ABI against float64 torch on the CPU: every prologue / epilogue
the one-frame launch chain uses, K split over 2 / 4 / 8 wave groups, several jobs in one launch, ragged sizes.
Tolerance: fp32 features within 1e-4 (BASELINE.json north_star) on O(1) activations."""
import numpy as np
import pytest
import torch

from ptt_amd import ops

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _layer(rs, cout, k):
    w = (rs.standard_normal((cout, k)) / np.sqrt(k)).astype(np.float32)
    return w, rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.standard_normal(cout).astype(np.float32)


@pytest.mark.parametrize("rows,K,Cout,act,res,cw", [(128, 256, 512, 1, True, 0), (128, 512, 256, 0, True, 1), (64, 256, 1536, 0, False, 0),
                                                    (2048, 512, 512, 1, False, 0), (1024, 512, 512, 0, False, 0),
                                                    (50, 40, 70, 1, True, 1), (50, 40, 70, 1, True, 2), (50, 40, 70, 1, True, 4),
                                                    (33, 259, 5, 0, False, 0), (130, 257, 256, 2, False, 0), (1, 8, 1, 0, False, 0)])
def test_plain_job(dev, rows, K, Cout, act, res, cw):
    rs = np.random.RandomState(rows * 7 + K + cw)
    x = rs.standard_normal((rows, K)).astype(np.float32)
    w, sc, sh = _layer(rs, Cout, K)
    r = rs.standard_normal((rows, Cout)).astype(np.float32) if res else None
    ref = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() * torch.from_numpy(sc).double() + torch.from_numpy(sh).double()
    raw_ref = ref.clone()
    ref = ref.clamp_min(0) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
    if res:
        ref = ref + torch.from_numpy(r).double()
    out = torch.full((rows, Cout), float('nan'), device=dev)
    raw = torch.full((rows, Cout), float('nan'), device=dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(w, dev)), Cout, x=_t(x, dev), scale=_t(sc, dev), shift=_t(sh, dev), act=act,
                              res=_t(r, dev) if res else None, out=out, raw=raw, col_tiles=cw)])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), **TOL)
    np.testing.assert_allclose(raw.cpu().numpy(), raw_ref.numpy(), **TOL)


def test_two_part_input_residual_and_output(dev):
    """vote_layer's last convolution as the head launches it (centroids_voting_head.py:86-94): input [feats | xyz] from two
    tensors, output columns 0-2 (+ xyz) to `votes`, columns 3.. (+ feats) into votes_feats[:, 1:] whose rows are 260 wide."""
    rs = np.random.RandomState(5)
    rows, C = 128, 256
    feats, xyz = rs.standard_normal((rows, C)).astype(np.float32), rs.standard_normal((rows, 3)).astype(np.float32)
    w, sc, sh = _layer(rs, C + 3, C + 3)
    x = np.concatenate([feats, xyz], 1)
    y = torch.from_numpy(x).double() @ torch.from_numpy(w).double().t() * torch.from_numpy(sc).double() + torch.from_numpy(sh).double()
    votes_ref = y[:, :3] + torch.from_numpy(xyz).double()
    vf_ref = y[:, 3:] + torch.from_numpy(feats).double()
    votes = torch.full((rows, 3), float('nan'), device=dev)
    vfeats = torch.full((rows, 260), -7.0, device=dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(w, dev)), C + 3, x=_t(feats, dev), x2=_t(xyz, dev), scale=_t(sc, dev), shift=_t(sh, dev),
                              res=_t(feats, dev), res2=_t(xyz, dev), res_split=3, out=vfeats, out_col0=1, out2=votes, out_split=3)])
    np.testing.assert_allclose(votes.cpu().numpy(), votes_ref.numpy(), **TOL)
    got = vfeats.cpu().numpy()
    np.testing.assert_allclose(got[:, 1:257], vf_ref.numpy(), **TOL)
    assert (got[:, 0] == -7.0).all() and (got[:, 257:] == -7.0).all()          # nothing else is touched


def _block_inputs(rs, B, N, D=512):
    P = B * N
    qkv = rs.standard_normal((P, 3 * D)).astype(np.float32)
    knn = np.stack([rs.permutation(N)[:16] for _ in range(P)]).astype(np.int32)
    pos = rs.standard_normal((P * 16, D)).astype(np.float32)
    return qkv, knn, pos


@pytest.mark.parametrize("B,N", [(1, 128), (1, 64), (3, 64), (2, 50)])
def test_delta_pair_and_aggregate_jobs(dev, B, N):
    """The three (point, neighbour)-row launches of a one-frame TransformerBlock (variants.py:158-163): fc_delta with its
    first Linear + ReLU formed in the A staging, fc_gamma[0] on q_i - k_j + pos_ij gathered in the A staging, fc_gamma[2]
    with the softmax over neighbours and the weighted sum as its epilogue."""
    rs = np.random.RandomState(B * 100 + N)
    D, P = 512, B * N
    qkv, knn, pos = _block_inputs(rs, B, N)
    rel = (rs.standard_normal((P * 16, 3)) * 0.5).astype(np.float32)
    w1 = (rs.standard_normal((D, 3)) / np.sqrt(3)).astype(np.float32)
    b1 = rs.standard_normal(D).astype(np.float32)
    w2, _, b2 = _layer(rs, D, D)
    wg1, _, bg1 = _layer(rs, D, D)
    wg2, _, bg2 = _layer(rs, D, D)
    d = lambda a: torch.from_numpy(a).double()
    # fc_delta
    pos_ref = (d(rel) @ d(w1).t() + d(b1)).clamp_min(0) @ d(w2).t() + d(b2)
    got_pos = torch.full((P * 16, D), float('nan'), device=dev)
    w1b = _t(np.concatenate([w1, b1[:, None]], 1), dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(w2, dev)), D, prologue=1, rel=_t(rel, dev), w1=w1b, K=D, shift=_t(b2, dev), out=got_pos)])
    np.testing.assert_allclose(got_pos.cpu().numpy(), pos_ref.numpy(), **TOL)
    # fc_gamma[0] on the pair input
    cloud = (np.arange(P) // N)[:, None] * N
    q, k, v = d(qkv[:, :D]), d(qkv[:, D:2 * D]), d(qkv[:, 2 * D:])
    t_ref = q[:, None, :] - k[torch.from_numpy(cloud + knn).long()] + d(pos).view(P, 16, D)
    g_ref = (t_ref.view(-1, D) @ d(wg1).t() + d(bg1)).clamp_min(0)
    got_g = torch.full((P * 16, D), float('nan'), device=dev)
    qkv_d, knn_d, pos_d = _t(qkv, dev), _t(knn, dev), _t(pos, dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(wg1, dev)), D, prologue=2, qkv=qkv_d, knn=knn_d, pos=pos_d, q_off=0, k_off=D, N=N,
                              K=D, shift=_t(bg1, dev), act=1, out=got_g)])
    np.testing.assert_allclose(got_g.cpu().numpy(), g_ref.numpy(), atol=2e-4, rtol=1e-4)
    # fc_gamma[2] + softmax over the neighbours + weighted sum
    a_ref = (g_ref @ d(wg2).t() + d(bg2)).view(P, 16, D)
    attn = torch.softmax(a_ref / np.sqrt(D), dim=1)
    res_ref = (attn * (v[torch.from_numpy(cloud + knn).long()] + d(pos).view(P, 16, D))).sum(1)
    got_res = torch.full((P, D), float('nan'), device=dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(wg2, dev)), D, x=_t(g_ref.float().numpy(), dev), epilogue=1, qkv=qkv_d, knn=knn_d,
                              pos=pos_d, v_off=2 * D, N=N, sm_scale=1.0 / np.sqrt(D), out=got_res)])
    np.testing.assert_allclose(got_res.cpu().numpy(), res_ref.numpy(), **TOL)


def test_jobs_of_one_launch_equal_their_own_launches(dev):
    """Four jobs of different shapes in ONE launch give bit-identical outputs to four launches."""
    rs = np.random.RandomState(11)
    shapes = [(128, 256, 1536, 0), (2048, 512, 512, 1), (128, 256, 1, 2), (77, 259, 259, 0)]
    data = []
    for rows, K, Cout, act in shapes:
        w, sc, sh = _layer(rs, Cout, K)
        data.append((_t(rs.standard_normal((rows, K)).astype(np.float32), dev), ops.pack_weight(_t(w, dev)), _t(sh, dev), Cout, act))
    alone = [torch.empty((x.shape[0], c), device=dev) for x, _, _, c, _ in data]
    for (x, wp, sh, c, act), o in zip(data, alone):
        ops.row_jobs([ops.row_job(wp, c, x=x, shift=sh, act=act, out=o)])
    together = [torch.empty_like(o) for o in alone]
    ops.row_jobs([ops.row_job(wp, c, x=x, shift=sh, act=act, out=o) for (x, wp, sh, c, act), o in zip(data, together)])
    for a, b in zip(alone, together):
        assert torch.equal(a, b)


def test_bad_jobs_are_refused(dev):
    x = torch.zeros((32, 64), device=dev)
    wp = ops.pack_weight(torch.zeros((32, 64), device=dev))
    out = torch.zeros((32, 32), device=dev)
    with pytest.raises(RuntimeError):
        ops.row_jobs([ops.row_job(wp, 32, x=x, out=out)] * 5)                       # more than PTT_ROW_JOBS_MAX
    with pytest.raises(RuntimeError):
        ops.row_jobs([ops.row_job(wp, 32, x=x, out=out, out_split=3)])              # a split without a second output


@pytest.mark.parametrize("B,N,M,ns,C", [(1, 128, 64, 16, 256), (2, 100, 33, 16, 256), (1, 256, 40, 32, 192)])
def test_sa_level_as_two_jobs(dev, B, N, M, ns, C):
    """A set-abstraction level with its first convolution hoisted (pointnet2_modules.py:57-90 on per-point terms): grouped
    layer 1 built in the A staging (term[idx] + Wx . rel, ReLU), layer 2 + max over the neighbours in the epilogue."""
    rs = np.random.RandomState(B * 1000 + M)
    xyz = rs.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    centres = xyz[:, :M].copy()
    idx = rs.randint(0, N, (B, M, ns)).astype(np.int32)
    term = rs.standard_normal((B, N, C)).astype(np.float32)
    wx = (rs.standard_normal((3, C)) * 0.5).astype(np.float32)
    w1, _, b1 = _layer(rs, 256, C)
    w2, _, b2 = _layer(rs, 256, 256)
    radius = 0.3
    d = lambda a: torch.from_numpy(a).double()
    bi = np.arange(B)[:, None, None]
    rel = (d(xyz)[bi, idx] - d(centres)[:, :, None, :]) / np.float32(radius)
    a = (d(term)[bi, idx] + rel @ d(wx)).clamp_min(0)                          # (B,M,ns,C)
    h = (a @ d(w1).t() + d(b1)).clamp_min(0)
    ref = (h @ d(w2).t() + d(b2)).max(dim=2)[0].clamp_min(0)                    # (B,M,256)
    hb = torch.empty((B * M * ns, 256), device=dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(w1, dev)), 256, prologue=3, x=_t(term, dev), idx=_t(idx, dev), xyz=_t(xyz, dev),
                              centres=_t(centres, dev), wx=_t(wx, dev), radius=radius, ns=ns, M=M, N=N, normalize_xyz=True,
                              pro_relu=True, shift=_t(b1, dev), act=1, out=hb)])
    np.testing.assert_allclose(hb.cpu().numpy(), h.reshape(-1, 256).numpy(), atol=2e-4, rtol=1e-4)
    out = torch.full((B, M, 256), float('nan'), device=dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(w2, dev)), 256, x=_t(h.reshape(-1, 256).float().numpy(), dev), epilogue=2, ns=ns, M=M,
                              shift=_t(b2, dev), act=1, out=out)])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), **TOL)


def test_tile_shape_does_not_change_a_bit(dev):
    """The K axis is always summed as the same eight-slice tree: col_tiles 1 / 2 / 4 (K split over 8 / 4 / 2 wave groups)
    give identical bits, so a frame's result does not depend on the batch it is evaluated in."""
    rs = np.random.RandomState(3)
    for rows, K, Cout in ((96, 512, 512), (40, 259, 70), (64, 256, 128)):
        w, sc, sh = _layer(rs, Cout, K)
        x, wp, shd = _t(rs.standard_normal((rows, K)).astype(np.float32), dev), ops.pack_weight(_t(w, dev)), _t(sh, dev)
        outs = []
        for cw in (1, 2, 4):
            o = torch.empty((rows, Cout), device=dev)
            ops.row_jobs([ops.row_job(wp, Cout, x=x, shift=shd, act=1, out=o, col_tiles=cw)])
            outs.append(o)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("B,N,npoints,k", [(1, 1024, (512, 256, 128), 16), (3, 512, (256, 128, 64), 0), (2, 300, (150, 100, 40), 16)])
def test_point_jobs_equal_the_level_by_level_launches(dev, B, N, npoints, k):
    """ptt_point_jobs_f32 (three ball-query levels + the seeds' kNN in one launch, every level read from the raw cloud through
    the level-0 sample) == ptt_centres_ball_query_f32 level by level + ptt_knn_rel_f32, bit for bit."""
    rs = np.random.RandomState(B + N)
    xyz = _t(rs.uniform(-1.5, 1.5, (B, N, 3)).astype(np.float32), dev)
    inds0 = ops.furthest_point_sampling(xyz, npoints[0])
    radii, ns = (0.3, 0.5, 0.7), (32, 32, 32)
    levels, inds64, knn = ops.sa_levels_point_jobs(xyz, inds0, list(npoints), radii, ns, knn_k=k)
    pts, sel = xyz, inds0
    for l in range(3):
        new_xyz, i64, idx = ops.centres_ball_query(pts, sel, npoints[l], radii[l], ns[l])
        assert torch.equal(levels[l][0], new_xyz) and torch.equal(levels[l][1], idx)
        if l == 0:
            assert torch.equal(inds64, i64)
        pts, sel = new_xyz, None
    if k:
        kidx, rel = ops.knn(pts, k, want_rel=True)
        assert torch.equal(knn[0], kidx) and torch.equal(knn[1], rel)


@pytest.mark.parametrize("B,N,M,k", [(1, 128, 64, 16), (3, 128, 64, 16), (2, 200, 50, 16), (1, 64, 64, 0), (2, 256, 128, 16)])
def test_fps_ball_knn_equals_the_three_launches(dev, B, N, M, k):
    """ptt_fps_ball_knn_f32 == ptt_fps_f32 + ptt_centres_ball_query_f32 + ptt_knn_rel_f32 on the centres, bit for bit — with
    duplicated points (exact distance ties) and points inside the 1e-3 origin ball in the cloud."""
    rs = np.random.RandomState(B * 31 + N)
    x = rs.uniform(-1.0, 1.0, (B, N, 3)).astype(np.float32)
    x[:, 5] = x[:, 3]; x[:, N // 2] = x[:, 1]; x[:, 7] = 0.001
    xyz = _t(x, dev)
    inds, inds64, new_xyz, idx, knn = ops.fps_ball_knn(xyz, M, 0.3, 16, k)
    ref_inds = ops.furthest_point_sampling(xyz, M)
    assert torch.equal(inds, ref_inds) and torch.equal(inds64, ref_inds.long())
    ref_xyz, _, ref_idx = ops.centres_ball_query(xyz, ref_inds, M, 0.3, 16)
    assert torch.equal(new_xyz, ref_xyz) and torch.equal(idx, ref_idx)
    if k:
        kidx, rel = ops.knn(ref_xyz, k, want_rel=True)
        assert torch.equal(knn[0], kidx) and torch.equal(knn[1], rel)


def test_xmax_operand(dev):
    rs = np.random.RandomState(9)
    a, b2 = rs.standard_normal((128, 256)).astype(np.float32), rs.standard_normal((128, 256)).astype(np.float32)
    w, sc, sh = _layer(rs, 256, 256)
    ref = torch.from_numpy(np.maximum(a, b2)).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(sh).double()
    out = torch.empty((128, 256), device=dev)
    both = _t(np.stack([a, b2]), dev)
    ops.row_jobs([ops.row_job(ops.pack_weight(_t(w, dev)), 256, x=both[0], xmax=both[1], shift=_t(sh, dev), out=out)])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), **TOL)
