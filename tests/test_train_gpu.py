"""N3 on the GPU: the hand-written training-step kernels (ptt_amd/csrc/train_ops.hip) against stock torch on the same
device, kernel by kernel, then the whole SharedMLP + pool autograd function (forward, every gradient, the BatchNorm
running statistics) against the reference's own op sequence in stock torch (pytorch_utils.SharedMLP in train mode +
max over the neighbour axis), and the full tracker's training step against fixture G10 (the REFERENCE model's loss and
parameter gradients)."""
import os

import numpy as np
import pytest
import torch

from ptt_amd import ops, train_ops
from ptt_amd.models.backbones_3d.pointnet2 import pytorch_utils as pt_utils
from tests.util import outside

# forward values are held against the north_star bar (1e-4) and the COUNT of elements outside it is asserted (measured on
# MI355X, printed by the tests): a handful of max-pool outputs whose arg-max row flips between two fp32 evaluations
Y_OUTSIDE_SA, Y_OUTSIDE_XCORR = 0, 14               # measured: 0 and 7 (of 98304)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("R,C", [(5000, 64), (2048 * 3 + 17, 131), (100, 7), (70000, 256)])
def test_bn_stats_apply_and_backward_kernels(dev, R, C):
    g = torch.Generator(device="cpu").manual_seed(R + C)
    z = (torch.randn(R, C, generator=g) * 2 + torch.randn(C, generator=g) * 3).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.2).to(dev)
    mean, var, invstd = ops.bn_stats(z, 1e-5)
    v64, m64 = torch.var_mean(z.double(), 0, unbiased=False)
    torch.testing.assert_close(mean.double(), m64, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(var.double(), v64, rtol=1e-5, atol=1e-7)
    act = ops.bn_apply(z, mean, invstd, gamma, beta, relu=True)
    ref = torch.relu((z.double() - m64) / torch.sqrt(v64 + 1e-5) * gamma.double() + beta.double())
    torch.testing.assert_close(act.double(), ref, rtol=1e-5, atol=1e-5)
    # backward against autograd through the same formula in float64
    zz = z.double().requires_grad_(True)
    gg, bb = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    v, m = torch.var_mean(zz, 0, unbiased=False)
    y = torch.relu((zz - m) / torch.sqrt(v + 1e-5) * gg + bb)
    up = torch.randn(R, C, generator=g).to(dev)
    y.backward(up.double())
    dz, dgamma, dbeta = ops.bn_bwd(up.clone(), act, z, mean, invstd, gamma)
    scale = float(zz.grad.abs().max())
    assert float((dz.double() - zz.grad).abs().max()) <= 2e-5 * scale + 1e-7
    torch.testing.assert_close(dgamma.double(), gg.grad, rtol=1e-4, atol=1e-3 * float(gg.grad.abs().max()) * 1e-1)
    torch.testing.assert_close(dbeta.double(), bb.grad, rtol=1e-4, atol=1e-3 * float(bb.grad.abs().max()) * 1e-1)
    # bit-reproducible
    dz2, dgamma2, _ = ops.bn_bwd(up.clone(), act, z, mean, invstd, gamma)
    assert torch.equal(dz, dz2) and torch.equal(dgamma, dgamma2)


@pytest.mark.parametrize("R,Cout,Cin", [(4096, 128, 128), (10000, 64, 3), (5000, 128, 131), (9000, 256, 259), (300, 5, 256),
                                         (40000, 512, 512),
                                         # at most 64 channels on one or both sides: the waves split the staged rows (SA0's layers)
                                         (50000, 64, 64), (33333, 128, 64), (20011, 64, 128), (7000, 64, 259), (6000, 260, 33),
                                         (300000, 64, 3),
                                         # the streaming form (>= 65536 rows, 64 inputs, 64 / 128 outputs): ragged chunk ends, odd rows
                                         (786432, 64, 64), (393216, 128, 64), (65537, 64, 64), (100001, 128, 64)])
def test_linear_wgrad_kernel(dev, R, Cout, Cin):
    g = torch.Generator(device="cpu").manual_seed(R)
    dz = torch.randn(R, Cout, generator=g).to(dev)
    x = torch.randn(R, Cin, generator=g).to(dev)
    got = ops.linear_wgrad(dz, x)
    ref = dz.double().t() @ x.double()
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-4 * np.sqrt(R) * 1e-2
    assert torch.equal(got, ops.linear_wgrad(dz, x))                     # fixed summation order
    acc = ops.linear_wgrad(dz, x, out=got.clone(), accumulate=True)
    torch.testing.assert_close(acc, 2 * got, rtol=1e-6, atol=1e-6)
    if Cin % 4 == 0 and R >= 20000:
        # the deferred activation of the layer input: x = relu(z * a + b) applied while the rows are staged, also on row views
        # of a wider buffer (row stride > channels)
        a, b = torch.randn(Cin, generator=g).to(dev), torch.randn(Cin, generator=g).to(dev)
        wide = torch.randn(R, Cin + 8, generator=g).to(dev)
        zv = wide[:, :Cin]
        got2 = ops.linear_wgrad(dz, zv, x_scale=a, x_shift=b)
        ref2 = dz.double().t() @ torch.relu(zv.double() * a.double() + b.double())
        assert float((got2.double() - ref2).abs().max()) <= 1e-5 * float(ref2.abs().max()) + 1e-4 * np.sqrt(R) * 1e-2


@pytest.mark.parametrize("G,ns,C", [(100, 32, 128), (77, 16, 256), (10, 64, 33)])
def test_pool_rows_kernels(dev, G, ns, C):
    x = torch.randn(G * ns, C, device=dev)
    x[0:ns, 0] = 1.5                                       # ties: the first row wins
    out, arg = ops.pool_rows(x, ns)
    ref, ridx = x.view(G, ns, C).max(dim=1)
    assert torch.equal(out, ref)
    assert int(arg[0, 0]) == 0
    assert torch.equal(torch.gather(x.view(G, ns, C), 1, arg.long()[:, None, :])[:, 0], ref)
    up = torch.randn(G, C, device=dev)
    dx = ops.pool_rows_bwd(up, arg, ns).view(G, ns, C)
    assert torch.equal(dx.sum(1), up) and int((dx != 0).sum()) <= G * C


@pytest.mark.parametrize("B,Cin,M,ns,spec,pool_dim", [(3, 3, 64, 32, [3, 64, 64, 128], 3), (2, 131, 40, 32, [131, 128, 128, 256], 3),
                                                       (2, 260, 64, 20, [260, 256, 256, 256], 2)])
def test_shared_mlp_pool_equals_stock_torch_training_step(dev, B, Cin, M, ns, spec, pool_dim):
    """Forward value, input gradient, all weight / BatchNorm gradients and the running statistics of the fused function
    against the reference op sequence (SharedMLP in train mode, then max over the pooled axis) in stock torch."""
    torch.manual_seed(5)
    a = pt_utils.SharedMLP(list(spec), bn=True).to(dev).train()
    b = pt_utils.SharedMLP(list(spec), bn=True).to(dev).train()
    b.load_state_dict(a.state_dict())
    with torch.no_grad():
        for u, v in zip(a, b):
            u.normlayer.bn.weight.uniform_(0.5, 1.5)
            u.normlayer.bn.bias.normal_(0, 0.2)
            v.normlayer.bn.weight.copy_(u.normlayer.bn.weight)
            v.normlayer.bn.bias.copy_(u.normlayer.bn.bias)
    shape = (B, Cin, M, ns) if pool_dim == 3 else (B, Cin, ns, M)
    x1 = torch.randn(*shape, device=dev, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    assert train_ops.usable(a, x1)
    y1 = train_ops.shared_mlp_pool(x1, a, pool_dim)
    y2 = b(x2).max(dim=pool_dim)[0]
    assert y1.shape == y2.shape
    torch.testing.assert_close(y1, y2, rtol=1e-4, atol=1e-4)
    up = torch.randn_like(y2)
    y1.backward(up)
    y2.backward(up)
    seen = {}

    def close(p, q, name, tol=3e-6):                      # measured: 1.2e-6 at worst
        err = float((p - q).abs().max()) / (float(q.abs().max()) + 1e-12)
        seen[name] = err
        assert err < tol, (name, err)

    close(x1.grad, x2.grad, "input grad")
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        close(p1.grad, p2.grad, n1)
    print("measured shared_mlp_pool %s: worst relative gradient error %.2e (%s)" % (spec, max(seen.values()), max(seen, key=seen.get)))
    for (n1, b1), (_, b2) in zip(a.named_buffers(), b.named_buffers()):
        torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-4, atol=1e-5, msg=n1)


def test_G10_training_step_on_the_row_kernels_matches_the_reference_gradients(dev):
    """Fixture G10 = loss and parameter gradients of the REFERENCE model (float32, CPU) for one seeded training step. The
    mirror in train mode on the GPU runs every stage on the hand-written row kernels; loss within 1e-5; every non-vanishing
    gradient's norm within 0.4 % and direction cos > 0.99997 (measured 0.21 % / 0.999989, printed)."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    g = np.load(os.path.join(GOLD, "G10_train_step.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    model = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g["seed"])).to(dev).train()
    ret, _, _ = model({'search_points': t(g["search"]), 'template_points': t(g["template"]), 'batch_size': 3,
                       'cls_label': t(g["cls_label"]), 'reg_label': t(g["reg_label"])})
    loss = ret['loss'].mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))             # measured 1.8e-6
    named = dict(model.named_parameters())
    keys = [str(k) for k in g["grad_keys"]]
    worst_norm, worst_cos = 0.0, 1.0
    per = {}
    for k, ref_norm in zip(keys, g["grad_norms"]):
        if ref_norm <= 1e-3:                                # mathematically zero (a bias in front of a softmax / BatchNorm)
            continue
        n = float(named[k].grad.double().norm())
        per[k] = abs(n - ref_norm) / ref_norm
        worst_norm = max(worst_norm, abs(n - ref_norm) / ref_norm)
    for k in sorted(per, key=per.get, reverse=True)[:8]:
        print("   G10 norm error %.4f  %s" % (per[k], k))
    for i, k in enumerate(str(k) for k in g["full_keys"]):
        ref = torch.from_numpy(g["grad_%d" % i]).double().flatten()
        if float(ref.norm()) <= 1e-3:
            continue
        got = named[k].grad.double().cpu().flatten()
        worst_cos = min(worst_cos, float(torch.dot(ref, got) / (ref.norm() * got.norm() + 1e-30)))
    above = sum(1 for v in per.values() if v > 1e-3)
    print("G10 on the row kernels: worst gradient-norm error %.4f, worst cosine %.6f, %d of %d gradients more than 0.1 %% off the "
          "reference's norm" % (worst_norm, worst_cos, above, len(per)))
    # measured on MI355X: 0.0021 / 0.999989 / 6 of 96 (round 2, before the BatchNorm variance of the fused statistics was made
    # cancellation-free: 0.022 / 0.9990; the reference's own float32 run is 0.0028 off its float64 gradient, fixture G14).
    # How much of that is rounding luck: two numerically innocent variants of this build measured 0.0054 (short launches on the
    # linear kernels) and 0.0075 (b = beta - mean * a as one FMA instead of a product and a difference) — every float32
    # evaluation of this network sits 0.2 - 0.8 % from the others; the bars below are twice THIS build's figures and a
    # change that moves them is not by itself a defect (compare against G14 before concluding anything)
    assert worst_norm < 0.004 and worst_cos > 0.99997 and above <= 12, (worst_norm, worst_cos, above)


def test_G15_training_step_at_configs3_sparsity_matches_the_reference_gradients(dev):
    """Fixture G15 = the reference tracker built from tools/cfgs/nuscenes_models/ptt.yaml, one training step on
    synthetic_train_batch(1515, 4) — K_s = 200 / K_t = 100 unique points, the sparsity BASELINE.json configs[3] trains at (G10 /
    G14 are KITTI-shaped). Same comparison as G10: loss, every non-vanishing gradient's norm, eight full gradients' direction."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from tests.util import fill_state_dict_
    g = np.load(os.path.join(GOLD, "G15_train_step_nuscenes.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    model = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g["seed"])).to(dev).train()
    ret, _, _ = model({'search_points': t(g["search"]), 'template_points': t(g["template"]), 'batch_size': int(g["batch"]),
                       'cls_label': t(g["cls_label"]), 'reg_label': t(g["reg_label"])})
    loss = ret['loss'].mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])), (float(loss.detach()), float(g["loss"]))
    named = dict(model.named_parameters())
    worst_norm, worst_cos, per = 0.0, 1.0, {}
    for k, ref_norm in zip((str(k) for k in g["grad_keys"]), g["grad_norms"]):
        if ref_norm <= 1e-3:
            continue
        per[k] = abs(float(named[k].grad.double().norm()) - ref_norm) / ref_norm
        worst_norm = max(worst_norm, per[k])
    for i, k in enumerate(str(k) for k in g["full_keys"]):
        ref = torch.from_numpy(g["grad_%d" % i]).double().flatten()
        if float(ref.norm()) <= 1e-3:
            continue
        got = named[k].grad.double().cpu().flatten()
        worst_cos = min(worst_cos, float(torch.dot(ref, got) / (ref.norm() * got.norm() + 1e-30)))
    above = sum(1 for v in per.values() if v > 1e-3)
    for k in sorted(per, key=per.get, reverse=True)[:5]:
        print("   G15 norm error %.4f  %s" % (per[k], k))
    print("G15 (K_s = 200) on the row kernels: worst gradient-norm error %.4f, worst cosine %.6f, %d of %d gradients more than 0.1 %% "
          "off the reference's norm" % (worst_norm, worst_cos, above, len(per)))
    # measured on MI355X: 0.0019 / 0.999971 / 12 of 97; bars at twice that, as for G10 (two float32 evaluations of this network sit
    # 0.2 - 0.8 % apart, see above)
    assert worst_norm < 0.004 and worst_cos > 0.99994 and above <= 24, (worst_norm, worst_cos, above)


def test_G14_float32_step_is_as_close_to_the_float64_gradient_as_the_references_float32_run(dev):
    """Fixture G14 = the reference's training step of G10 in FLOAT64 (tests/golden/make_golden_f64.py). The tracker's max-pools
    and ReLUs make per-cent differences between two float32 evaluations possible, so the yardstick is the float64 gradient: this
    build's float32 step on the row kernels must be no further from it than twice what the reference's own float32 run (G10)
    is — measured: 0.0019 (ours) against 0.0028 (reference float32) in the worst gradient norm — and must pick the reference's
    64 proposals out of the 128 votes (the one data-dependent sampling of the step)."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from ptt_amd.models.backbones_3d.pointnet2 import pointnet2_utils as PU
    from tests.util import fill_state_dict_
    g = np.load(os.path.join(GOLD, "G10_train_step.npz"))
    g14 = np.load(os.path.join(GOLD, "G14_train_step_f64.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    model = fill_state_dict_(build_network(ptt_model_cfg(), 1, StubDataset(training=True)), int(g["seed"])).to(dev).train()
    picks, fps = [], PU.furthest_point_sample

    def spy(xyz, npoint):
        out = fps(xyz, npoint)
        if xyz.shape[1] == 128 and npoint == 64:
            picks.append(out.cpu().numpy())
        return out
    PU.furthest_point_sample = spy
    try:
        ret, _, _ = model({'search_points': t(g["search"]), 'template_points': t(g["template"]), 'batch_size': 3,
                           'cls_label': t(g["cls_label"]), 'reg_label': t(g["reg_label"])})
    finally:
        PU.furthest_point_sample = fps
    loss = ret['loss'].mean()
    loss.backward()
    assert len(picks) == 1 and [set(r) for r in picks[0].tolist()] == [set(r) for r in g14["vote_picks"].tolist()]
    assert abs(float(loss.detach()) - float(g14["loss"])) <= 5e-6 * abs(float(g14["loss"]))          # measured 1.4e-6
    named = dict(model.named_parameters())
    keys = [str(k) for k in g14["grad_keys"]]
    n64 = dict(zip(keys, g14["grad_norms"]))
    n32 = dict(zip([str(k) for k in g["grad_keys"]], g["grad_norms"]))
    ours = max(abs(float(named[k].grad.double().norm()) - n64[k]) / n64[k] for k in keys if n64[k] > 1e-3)
    ref = max(abs(n32[k] - n64[k]) / n64[k] for k in keys if n64[k] > 1e-3)
    l2 = 0.0
    for i, k in enumerate(str(k) for k in g14["full_keys"]):
        r64 = torch.from_numpy(g14["grad_%d" % i]).double().flatten()
        if float(r64.norm()) > 1e-3:
            l2 = max(l2, float((named[k].grad.double().cpu().flatten() - r64).norm() / r64.norm()))
    print("G14: worst gradient-norm error against float64 — this build (float32) %.4f, the reference's float32 run %.4f; worst "
          "relative L2 error of a full gradient %.4f" % (ours, ref, l2))
    assert ours <= 2.0 * ref and l2 <= 0.011, (ours, ref, l2)                                           # measured 0.0019 / 0.0028 / 0.0054


def _clone_module(m):
    import copy
    return copy.deepcopy(m)


@pytest.mark.parametrize("B,N,M,C,spec,radius,ns", [(3, 256, 128, 128, [128, 128, 128, 256], 0.5, 32),
                                                     (2, 128, 64, 257, [257, 256, 256, 256], 0.3, 16)])
def test_hoisted_sa_level_training_equals_the_reference_op_sequence(dev, B, N, M, C, spec, radius, ns):
    """PointnetSAModuleVotes in train mode: the hoisted row-kernel path (layer 0 once per point, gather rows, row
    kernels) against the reference op sequence (QueryAndGroup -> SharedMLP -> max) in stock torch: output, gradients
    with respect to the coordinates (the box head's votes need them), the features and every parameter."""
    from ptt_amd import synth
    from ptt_amd.models.backbones_3d.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(11)
    a = PointnetSAModuleVotes(mlp=list(spec), radius=radius, nsample=ns, use_xyz=True, normalize_xyz=True,
                              sample_method='fps').to(dev).train()
    b = _clone_module(a)
    s, _ = synth.frames(9, B, N, 64, K_s=N // 2)
    xyz1 = torch.from_numpy(s).to(dev).requires_grad_(True)
    f1 = torch.randn(B, C, N, device=dev, requires_grad=True)
    xyz2, f2 = xyz1.detach().clone().requires_grad_(True), f1.detach().clone().requires_grad_(True)
    nx1, y1, i1 = a(xyz1, f1, M)
    orig = train_ops.usable
    train_ops.usable = lambda *k: False                     # the reference op sequence on stock layers
    try:
        nx2, y2, i2 = b(xyz2, f2, M)
    finally:
        train_ops.usable = orig
    assert torch.equal(i1, i2) and torch.equal(nx1, nx2)
    n_out, worst_y = outside(y1, y2)
    print("measured hoisted SA level %s: %d of %d outputs outside 1e-4, worst %.2f x the tolerance" % (spec, n_out, y2.numel(), worst_y))
    assert n_out <= Y_OUTSIDE_SA and worst_y <= 0.1, (n_out, worst_y)      # measured: 0.03 x the tolerance
    up = torch.randn_like(y2)
    (y1 * up).sum().backward()
    (y2 * up).sum().backward()
    seen = {}

    def close(p, q, name, tol=3e-5):                      # measured: 6.8e-6 alone (six runs, the same digits every time); 1.7e-5 once inside
        #                                                   the whole suite, in a weight gradient: the STOCK side picks its convolution-
        #                                                   backward algorithm by what the process ran before (this side's kernels do not)
        err = float((p - q).abs().max()) / (float(q.abs().max()) + 1e-12)
        seen[name] = err
        assert err < tol, (name, err)

    close(f1.grad, f2.grad, "feature grad")
    close(xyz1.grad, xyz2.grad, "xyz grad")
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        close(p1.grad, p2.grad, n1)
    print("measured hoisted SA level %s: worst relative gradient error %.2e (%s)" % (spec, max(seen.values()), max(seen, key=seen.get)))
    for (n1, b1), (_, b2) in zip(a.named_buffers(), b.named_buffers()):
        torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-3, atol=1e-5, msg=n1)


@pytest.mark.parametrize("B,N,M,C,spec,radius,ns,method", [(3, 512, 256, 0, [0, 64, 64, 128], 0.3, 32, 'fps'),
                                                            (2, 256, 128, 128, [128, 128, 128, 256], 0.5, 32, 'sequence'),
                                                            (2, 200, 100, 256, [256, 128, 128, 256], 0.7, 32, 'fps')])
def test_sa_level_with_fixed_coordinates_one_launch_front_equals_the_reference_op_sequence(dev, B, N, M, C, spec, radius, ns, method):
    """The backbone's levels in train mode (coordinates carry no gradient): centres + ball query in one launch, layer 0 per
    (centre, neighbour) row in one more (ptt_sa_z0_rows_f32) — with point features (their half hoisted per point) and
    without (SA0: the first convolution is the three coordinate channels) — against the reference op sequence."""
    from ptt_amd import synth
    from ptt_amd.models.backbones_3d.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(12)
    a = PointnetSAModuleVotes(mlp=list(spec), radius=radius, nsample=ns, use_xyz=True, normalize_xyz=True,
                              sample_method=method).to(dev).train()
    b = _clone_module(a)
    s, _ = synth.frames(10, B, N, 64, K_s=N // 2)
    xyz = torch.from_numpy(s).to(dev)
    f1 = torch.randn(B, C, N, device=dev, requires_grad=True) if C else None
    f2 = f1.detach().clone().requires_grad_(True) if C else None
    calls = {"z0": 0}
    real = ops.sa_z0_rows

    def counting(*k, **kw):
        calls["z0"] += 1
        return real(*k, **kw)

    ops.sa_z0_rows = counting
    try:
        nx1, y1, i1 = a(xyz, f1, M)
    finally:
        ops.sa_z0_rows = real
    assert calls["z0"] == 1                                  # the one-launch front ran
    orig = train_ops.usable
    train_ops.usable = lambda *k: False                     # the reference op sequence on stock layers
    try:
        nx2, y2, i2 = b(xyz, f2, M)
    finally:
        train_ops.usable = orig
    assert torch.equal(i1, i2) and i1.dtype == torch.int64 and torch.equal(nx1, nx2)
    n_out, worst_y = outside(y1, y2)
    assert n_out <= Y_OUTSIDE_SA and worst_y <= 0.1, (n_out, worst_y)
    up = torch.randn_like(y2)
    (y1 * up).sum().backward()
    (y2 * up).sum().backward()
    worst = 0.0
    pairs = [(p1.grad, p2.grad, n1) for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters())]
    if C:
        pairs.append((f1.grad, f2.grad, "feature grad"))
    for p, q, name in pairs:
        err = float((p - q).abs().max()) / (float(q.abs().max()) + 1e-12)
        worst = max(worst, err)
        assert err < 5e-5, (name, err)                     # measured 2.5e-5 (the K = 3 level's first BatchNorm bias), 7e-6 elsewhere
    print("measured one-launch SA front %s: worst relative gradient error %.2e" % (spec, worst))
    for (n1, b1), (_, b2) in zip(a.named_buffers(), b.named_buffers()):
        torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-3, atol=1e-5, msg=n1)


def test_hoisted_cosine_sim_aug_training_equals_the_reference_op_sequence(dev):
    from ptt_amd.hot_path import AttrDict
    from ptt_amd.models.similarity_modules.p2b_xcoor import CosineSimAug
    torch.manual_seed(3)
    cfg = AttrDict.wrap(dict(DEBUG=False, MLP=dict(CHANNELS=[260, 256, 256, 256], BN=True), CONV=dict(CHANNELS=[256, 256, 256], BN=True)))
    a = CosineSimAug(cfg).to(dev).train()
    b = _clone_module(a)
    B = 3
    sf1 = torch.randn(B, 256, 128, device=dev, requires_grad=True)
    tf1 = torch.randn(B, 256, 64, device=dev, requires_grad=True)
    tx1 = (torch.rand(B, 64, 3, device=dev) * 4 - 2).requires_grad_(True)
    sf2, tf2, tx2 = (t.detach().clone().requires_grad_(True) for t in (sf1, tf1, tx1))
    y1 = a({'search_feats': sf1, 'template_feats': tf1, 'template_seeds': tx1})['cosine_feats']
    orig = train_ops.usable
    train_ops.usable = lambda *k: False
    try:
        y2 = b({'search_feats': sf2, 'template_feats': tf2, 'template_seeds': tx2})['cosine_feats']
    finally:
        train_ops.usable = orig
    n_out, worst_y = outside(y1, y2)
    print("measured hoisted CosineSimAug: %d of %d outputs outside 1e-4, worst %.2f x the tolerance" % (n_out, y2.numel(), worst_y))
    assert n_out <= Y_OUTSIDE_XCORR and worst_y <= 2.0, (n_out, worst_y)   # measured: 7 outside, worst 1.09 x (the stock side's
                                                                         # float32 BatchNorm variance is the less accurate one)
    up = torch.randn_like(y2)
    (y1 * up).sum().backward()
    (y2 * up).sum().backward()
    # the search features reach the output through ONE of the 260 fusion channels (the cosine), so their gradient is a
    # small sum of max-pool routes and a handful of routes that flip between two fp32 evaluations of z0 (1e-6 apart)
    # shows: L2-relative 2e-2 there, 1e-2 of the maximum everywhere else (observed 4e-3)
    err = float((sf1.grad - sf2.grad).norm() / sf2.grad.norm())
    print("measured hoisted CosineSimAug: search-feature gradient L2-relative error %.2e" % err)
    assert err < 4e-4, ("search grad", err)               # measured 1.6e-4
    for p, q, name in ((tf1.grad, tf2.grad, "template grad"), (tx1.grad, tx2.grad, "xyz grad")):
        err = float((p - q).abs().max()) / (float(q.abs().max()) + 1e-12)
        print("measured hoisted CosineSimAug: %s error %.2e of the maximum" % (name, err))
        assert err < 1e-3, (name, err)                     # measured 4.5e-4
    # scale: a gradient that is mathematically zero (mlp.layer2's BatchNorm bias: a per-channel shift in front of the
    # train-mode BatchNorm of conv[0]) is rounding noise in both implementations — errors are measured against
    # max(|reference gradient|, 1e-3 of the largest parameter gradient)
    gmax = max(float(p.grad.abs().max()) for p in b.parameters())
    worst_p = 0.0
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        err = float((p1.grad - p2.grad).abs().max()) / max(float(p2.grad.abs().max()), 1e-3 * gmax)
        worst_p = max(worst_p, err)
        assert err < 2.5e-3, (n1, err)            # measured: up to 1.1e-3 (max-pool routes that flip between the two evaluations)
    print("measured hoisted CosineSimAug: worst parameter-gradient error %.2e" % worst_p)


@pytest.mark.parametrize("B,N,E,C", [(3, 256, 128 * 32, 128), (2, 100, 37, 8), (1, 16, 16384, 64)])
def test_gather_rows_and_its_deterministic_adjoint(dev, B, N, E, C):
    g = torch.Generator(device="cpu").manual_seed(E)
    rows = torch.randn(B, N, C, generator=g).to(dev)
    idx = torch.randint(0, max(1, N // 3), (B, E), generator=g).to(torch.int32).to(dev)
    out = ops.gather_rows(rows, idx)
    assert torch.equal(out, torch.gather(rows, 1, idx.long()[..., None].expand(-1, -1, C)))
    up = torch.randn(B, E, C, generator=g).to(dev)
    adj = ops.scatter_rows_det(up, idx, N)
    ref = torch.zeros(B, N, C, dtype=torch.float64, device=dev).index_put_(
        (torch.arange(B, device=dev)[:, None].expand(B, E), idx.long()), up.double(), accumulate=True)
    assert float((adj.double() - ref).abs().max()) <= 1e-5 * (float(ref.abs().max()) + 1)
    assert torch.equal(adj, ops.scatter_rows_det(up, idx, N))


@pytest.mark.parametrize("N,xyz_grad", [(128, False), (64, True)])
def test_transformer_block_training_path_equals_the_reference_op_sequence(dev, N, xyz_grad):
    """TransformerBlock in train mode: the hand-written element-wise passes (pair input, softmax-over-neighbours +
    weighted sum, their backward with deterministic neighbour scatter-adds) against the reference's op sequence in stock
    torch (variants.py:149-165): res, attn, the gradients of the features, of the coordinates (the box head's proposal
    centres carry gradient) and of every parameter."""
    import copy
    from ptt_amd.models.transformer_block.variants import TransformerBlock
    torch.manual_seed(N)
    a = TransformerBlock(256, 512, 16).to(dev).train()
    b = copy.deepcopy(a)
    B = 3
    xyz1 = (torch.rand(B, N, 3, device=dev) * 4 - 2).requires_grad_(xyz_grad)
    f1 = torch.randn(B, N, 256, device=dev, requires_grad=True)
    xyz2, f2 = xyz1.detach().clone().requires_grad_(xyz_grad), f1.detach().clone().requires_grad_(True)
    assert train_ops.pt_block_usable(a, xyz1, f1)
    r1, at1 = a(xyz1, f1)
    orig = train_ops.pt_block_usable
    train_ops.pt_block_usable = lambda *k: False
    try:
        r2, at2 = b(xyz2, f2)
    finally:
        train_ops.pt_block_usable = orig
    torch.testing.assert_close(r1, r2, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(at1, at2, rtol=1e-4, atol=1e-5)
    up = torch.randn_like(r2)
    (r1 * up).sum().backward()
    (r2 * up).sum().backward()

    seen = {}

    def close(p, q, name, tol=8e-6):                      # measured: 3.3e-6 at worst
        err = float((p - q).abs().max()) / max(float(q.abs().max()), 1e-6)
        seen[name] = err
        assert err < tol, (name, err)

    close(f1.grad, f2.grad, "feature grad")
    if xyz_grad:
        close(xyz1.grad, xyz2.grad, "xyz grad")
    gmax = max(float(p.grad.abs().max()) for p in b.parameters())
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        err = float((p1.grad - p2.grad).abs().max()) / max(float(p2.grad.abs().max()), 1e-3 * gmax)   # fc_gamma.2.bias: exactly 0
        seen[n1] = err
        assert err < 8e-6, (n1, err)
    print("measured transformer block (training, N = %d): worst relative gradient error %.2e (%s)" % (N, max(seen.values()), max(seen, key=seen.get)))


@pytest.mark.parametrize("G,ns,C", [(300, 32, 128), (77, 16, 256), (5000, 32, 64)])
def test_pooled_bn_backward_equals_pool_backward_then_dense_bn_backward(dev, G, ns, C):
    """ptt_bn_bwd_pooled_f32 (max-pool backward + BatchNorm/ReLU backward from the POOLED gradient; the (R, C) gradient is
    never materialised) against the two-step form ptt_pool_rows_bwd_f32 -> ptt_bn_bwd_f32 it replaces: dz to rounding (the
    sums run over the same non-zero terms in a different grouping), bit-reproducible run to run."""
    g = torch.Generator(device="cpu").manual_seed(G + C)
    R = G * ns
    z = (torch.randn(R, C, generator=g) * 2 + torch.randn(C, generator=g)).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.2).to(dev)
    mean, var, invstd = ops.bn_stats(z, 1e-5)
    a = (gamma * invstd).contiguous()
    b = (beta - mean * a).contiguous()
    pooled, arg = ops.pool_rows(z, ns, a, b)
    up = torch.randn(G, C, generator=g).to(dev)
    dense = ops.pool_rows_bwd(up, arg, ns)
    dz_ref, dgamma_ref, dbeta_ref = ops.bn_bwd(dense, None, z, mean, invstd, gamma, act_scale=a, act_shift=b)
    dz, dgamma, dbeta = ops.bn_bwd_pooled(up, arg, ns, z, mean, invstd, gamma, a, b)
    scale = float(dz_ref.abs().max())
    assert float((dz - dz_ref).abs().max()) <= 2e-6 * scale
    torch.testing.assert_close(dgamma, dgamma_ref, rtol=1e-5, atol=1e-5 * float(dgamma_ref.abs().max()))
    torch.testing.assert_close(dbeta, dbeta_ref, rtol=1e-5, atol=1e-5 * float(dbeta_ref.abs().max()))
    dz2, dgamma2, _ = ops.bn_bwd_pooled(up, arg, ns, z, mean, invstd, gamma, a, b)
    assert torch.equal(dz, dz2) and torch.equal(dgamma, dgamma2)
    # the SyncBatchNorm split gives the same numbers from the same sums
    sums = ops.bn_bwd_pooled_sums(up, arg, ns, z, mean, invstd, a, b).float()
    count = torch.full((1,), float(R), dtype=torch.float64, device=dev)
    dz3 = ops.bn_bwd_pooled_apply(up, arg, ns, z, mean, invstd, gamma, sums[0].contiguous(), sums[1].contiguous(), count, a, b)
    assert float((dz3 - dz).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("channels,residual", [([256, 256, 256, 1], False), ([259, 256, 256, 259], True), ([256, 256, 256, 5], False)])
def test_conv1d_stack_training_rows_path_equals_stock_torch(dev, channels, residual):
    """The heads' Conv1d stacks ([Conv1d + BatchNorm1d + ReLU] x 2 + Conv1d with bias; centroids_voting_head.py:15-21,
    box_voting_head.py:25) in TRAINING mode on the row kernels against the same modules in stock torch on (B,C,N) tensors:
    output, input gradient, every parameter gradient, the running statistics."""
    import copy
    torch.manual_seed(7)
    a = (pt_utils.Seq(channels[0]).conv1d(channels[1], bn=True).conv1d(channels[2], bn=True).conv1d(channels[3], activation=None)).to(dev).train()
    with torch.no_grad():
        for u in a:
            if hasattr(u, 'normlayer'):
                u.normlayer.bn.weight.uniform_(0.5, 1.5)
                u.normlayer.bn.bias.normal_(0, 0.2)
    b = copy.deepcopy(a)
    B, N = 4, 128
    x1 = torch.randn(B, N, channels[0], device=dev, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    assert train_ops.conv1d_stack_usable(a, x1)
    y1 = train_ops.conv1d_stack_rows(a, x1, residual=x1 if residual else None)
    y2 = b(x2.transpose(1, 2)).transpose(1, 2)
    if residual:
        y2 = y2 + x2
    n_out, worst = outside(y1, y2)
    assert n_out == 0, (n_out, worst)
    up = torch.randn_like(y2)
    (y1 * up).sum().backward()
    (y2 * up).sum().backward()
    seen = {"input": float((x1.grad - x2.grad).abs().max() / x2.grad.abs().max())}
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        seen[n1] = float((p1.grad - p2.grad).abs().max()) / max(float(p2.grad.abs().max()), 1e-6)
    print("measured conv1d stack %s: worst relative gradient error %.2e (%s)" % (channels, max(seen.values()), max(seen, key=seen.get)))
    assert max(seen.values()) < 2e-6, seen                 # measured: 5.7e-7
    for (n1, b1), (_, b2) in zip(a.named_buffers(), b.named_buffers()):
        torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-4, atol=1e-5, msg=n1)
