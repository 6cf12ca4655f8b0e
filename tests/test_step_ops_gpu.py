"""The ends of the training step on the GPU (ptt_amd/csrc/step_ops.hip): the one-launch tracking losses against the heads' own
stock-torch losses (the mirror of reference centroids_voting_head.py:29-62 / box_voting_head.py:33-66,96-104) — values and every
gradient — and ClipAdam against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (tools/train_utils/train_utils.py:47-51)."""
import numpy as np
import pytest
import torch

from ptt_amd import ops, train_ops
from ptt_amd.optim import ClipAdam

pytestmark = pytest.mark.gpu


def _loss_inputs(dev, B, N, Ns, M, seed, near):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    reg = torch.cat((r(B, 3) * 0.5, r(B, 1)), dim=1)
    votes = reg[:, None, :3] + r(B, N, 3) * 0.8                          # residuals on both sides of smooth-L1's knee
    centres = reg[:, None, :3] + r(B, M, 3) * near                       # proposals inside 0.3, between, and beyond 0.6
    box = torch.cat((centres + r(B, M, 3) * 0.7, reg[:, None, 3:] + r(B, M, 1) * 1.5, r(B, M, 1) * 3), dim=2)
    cls_points = (torch.rand(B, Ns, generator=g) > 0.6).float()
    inds = torch.stack([torch.randperm(Ns, generator=g)[:N] for _ in range(B)])
    t = [r(B, N) * 3, votes, box, centres, cls_points, inds, reg]
    return [x.to(dev).contiguous() for x in t]


def _stock_losses(seed_cls, votes, box, centres, cls_points, inds, reg, pw_s, pw_b, w):
    """The heads' op sequences (ptt_amd/models/voting_heads/*.py get_*_loss, _proposal_labels) written out."""
    bce = lambda pw, red: torch.nn.BCEWithLogitsLoss(pos_weight=pw, reduction=red)
    sl1 = torch.nn.SmoothL1Loss(reduction='none')
    label = cls_points.gather(1, inds)
    l1 = bce(pw_s, 'mean')(seed_cls.view(-1), label.view(-1))
    l2 = (sl1(votes, reg[:, None, :3].expand_as(votes)).mean(2) * label).sum() / (label.sum() + 1e-06)
    dist = torch.sqrt(torch.sum((centres - reg[:, None, 0:3]) ** 2, dim=-1) + 1e-6)
    y = torch.zeros_like(dist); m = torch.zeros_like(dist)
    y[dist < 0.3] = 1; m[dist < 0.3] = 1; m[dist > 0.6] = 1
    l3 = torch.sum(bce(pw_b, 'none')(box[:, :, -1], y) * m) / (torch.sum(m) + 1e-6)
    pred = box[:, :, :-1]
    l4 = (sl1(pred, reg[:, None, :].expand_as(pred)).mean(2) * y).sum() / (y.sum() + 1e-06)
    total = (l1.float() * w[0] + l2.float() * w[1]).float() + (l3.float() * w[2] + l4.float() * w[3]).float()
    return total, (l1, l2, l3, l4), (y, m)


@pytest.mark.parametrize("B,N,Ns,M,near", [(48, 128, 1024, 64, 0.3), (3, 128, 1024, 64, 0.25), (2, 37, 50, 5, 0.4), (4, 16, 16, 8, 5.0)])
def test_one_launch_losses_equal_the_heads_stock_losses(dev, B, N, Ns, M, near):
    """near = 5.0: no proposal within 0.3 of the box centre — the label sum is zero and the box regression loss 0 / 1e-6 = 0."""
    seed_cls, votes, box, centres, cls_points, inds, reg = _loss_inputs(dev, B, N, Ns, M, 11 + B, near)
    pw_s, pw_b = torch.tensor([1.0], device=dev), torch.tensor([2.0], device=dev)
    w = (0.2, 1.0, 1.5, 0.2)
    leaves = [t.clone().requires_grad_(True) for t in (seed_cls, votes, box)]
    ref_total, ref_parts, (y, m) = _stock_losses(leaves[0], leaves[1], leaves[2], centres, cls_points, inds, reg, pw_s, pw_b, w)
    (ref_total * 0.75).backward()
    mine = [t.clone().requires_grad_(True) for t in (seed_cls, votes, box)]
    total, vals = train_ops.track_losses(mine[0], mine[1], mine[2], centres, cls_points, inds, reg, pw_s, pw_b, w)
    (total * 0.75).backward()
    assert abs(float(total.detach()) - float(ref_total.detach())) <= 2e-6 * max(1.0, abs(float(ref_total.detach())))
    for k, r in enumerate(ref_parts):
        assert abs(float(vals[1 + k]) - float(r.detach())) <= 2e-6 * max(1.0, abs(float(r.detach()))), k
    out = vals.device_values
    assert float(out[5]) == float(cls_points.gather(1, inds).sum()) and float(out[6]) == float(m.sum()) and float(out[7]) == float(y.sum())
    for a, b in zip(mine, leaves):
        scale = float(b.grad.abs().max()) + 1e-12
        assert float((a.grad - b.grad).abs().max()) <= 2e-6 * scale + 1e-10
    # labels given per seed (no index list)
    t2, v2 = train_ops.track_losses(seed_cls, votes, box, centres, cls_points.gather(1, inds).contiguous(), None, reg, pw_s, pw_b, w)
    assert float(t2.detach()) == float(total.detach())


def test_loss_values_are_fetched_once_and_only_when_read(dev):
    dv = torch.arange(8, dtype=torch.float32, device=dev)
    vals = train_ops.LossValues(dv)
    a, b = vals[1], vals[3]
    assert vals.host is None                                        # nothing copied yet
    assert float(a) == 1.0 and vals.host is not None
    assert "%.1f" % b == "3.0" and a + b == 4.0 and b.item() == 3.0 and repr(a) == "1.0" and b > a


def test_full_model_training_loss_takes_the_one_launch_path_and_matches_the_stock_losses(dev):
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from ptt_amd.train_step import synthetic_train_batch
    torch.manual_seed(3)
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    batch = synthetic_train_batch(5, 4, dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ret, tb, disp = model(dict(batch))
    assert isinstance(tb['centroids_cls_loss'], train_ops.LossValues.Value) and set(tb) == set(disp) == {
        'centroids_cls_loss', 'centroids_reg_loss', 'boxes_cls_loss', 'boxes_reg_loss'}
    ret['loss'].mean().backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    model._one_launch_losses = lambda: None                          # the heads' own losses (the mirror of the reference's)
    ret2, tb2, _ = model(dict(batch))
    assert isinstance(tb2['centroids_cls_loss'], float)
    ret2['loss'].mean().backward()
    assert abs(float(ret['loss'].detach()) - float(ret2['loss'].detach())) <= 2e-6 * abs(float(ret2['loss'].detach()))
    for k in tb2:
        assert abs(float(tb[k]) - tb2[k]) <= 2e-6 * max(1.0, abs(tb2[k])), k
    named = dict(model.named_parameters())
    assert set(g1) == {k for k, p in named.items() if p.grad is not None}
    for k, g in g1.items():
        ref = named[k].grad
        assert float((g - ref).norm()) <= 2e-5 * float(ref.norm()) + 1e-9, k


@pytest.mark.parametrize("max_norm,scale", [(10.0, 1.0), (10.0, 300.0), (None, 1.0)])
def test_clip_adam_equals_clip_grad_norm_then_torch_adam(dev, max_norm, scale):
    """Three steps on the same gradients: parameters, both moments, the clipped .grad and the reported norm; scale = 300: the
    clip is active. Different lr / step counts exercise the bias corrections; weight_decay = 0.01 the L2 term."""
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 67), (4100,), (1,), (3, 5, 7), (256, 256), (9000,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    g = torch.Generator().manual_seed(5); pa = mk()
    g = torch.Generator().manual_seed(5); pb = mk()
    kw = dict(lr=1e-3, betas=(0.5, 0.999), eps=1e-6, weight_decay=0.01)
    oa, ob = ClipAdam(pa, **kw), torch.optim.Adam(pb, foreach=True, **kw)
    gg = torch.Generator().manual_seed(9)
    gmax = {}
    for step in range(3):
        grads = [torch.randn(*s, generator=gg).to(dev) * scale * (1 + step) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(max_norm=max_norm)
        if max_norm is not None:
            norm = torch.nn.utils.clip_grad_norm_(pb, max_norm)
            assert abs(float(oa.last_norm) - float(norm)) <= 2e-6 * float(norm)
        ob.step()
        for p, q in zip(pa, pb):
            assert float((p - q).abs().max()) <= 1e-6 * float(q.abs().max()) + 2e-9, step
            assert float((p.grad - q.grad).abs().max()) <= 2e-6 * float(q.grad.abs().max()) + 1e-12
            for key in ('exp_avg', 'exp_avg_sq'):
                a, b = oa.state[p][key], ob.state[q][key]
                # a moment of a single element can cancel: the bar is relative to the gradients that went into it
                gmax[id(q)] = max(gmax.get(id(q), 0.0), float(q.grad.abs().max()))
                ref_scale = max(float(b.abs().max()), gmax[id(q)] ** (2 if key == 'exp_avg_sq' else 1))
                assert float((a - b).abs().max()) <= 2e-6 * ref_scale + 1e-12, (step, key)
            assert float(oa.state[p]['step']) == float(ob.state[q]['step']) == step + 1
    # the state dict of one loads into the other
    ob.load_state_dict(oa.state_dict())
    oa.load_state_dict(ob.state_dict())


def test_clip_adam_takes_the_stock_path_for_what_the_table_does_not_hold(dev):
    p_gpu = torch.nn.Parameter(torch.ones(10, device=dev))
    p_half = torch.nn.Parameter(torch.ones(10, device=dev, dtype=torch.float64))
    o = ClipAdam([p_gpu, p_half], lr=0.1)
    p_gpu.grad, p_half.grad = torch.ones_like(p_gpu), torch.ones_like(p_half)
    o.step(max_norm=1.0)
    ref_g = 1.0 / (20 ** 0.5)
    assert abs(float(p_gpu.grad[0]) - ref_g) < 1e-6 and abs(float(p_gpu[0]) - 0.9) < 1e-5 and abs(float(p_half[0]) - 0.9) < 1e-5


@pytest.mark.parametrize("R,C,pad", [(6144, 259, 0), (98304, 512, 0), (300, 3, 0), (70000, 64, 4), (5000, 1, 0), (128, 256, 0), (9999, 37, 3)])
def test_colsum_is_exact_to_float64_rounding_and_bit_reproducible(dev, R, C, pad):
    """The bias gradients of the row-wise layers (ptt_colsum_f32): float4 path and the scalar path (259 = 3 + 256 channels of the
    vote layer, row views of a wider buffer), one chunk and many."""
    g = torch.Generator().manual_seed(R + C)
    wide = torch.randn(R, C + pad, generator=g).to(dev)
    x = wide[:, :C] if pad else wide
    got = ops.colsum(x)
    ref = x.double().sum(0)
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(x.abs().double().sum(0).max())
    assert torch.equal(got, ops.colsum(x))


def test_cos_map_function_equals_the_elementwise_formulation(dev):
    """train_ops._CosMap against x1 . x2 / (max(|x1|, eps) max(|x2|, eps)) written with torch ops — values and both gradients, for
    channel-major inputs and (B,C,n) views of point-major storage, with one all-zero feature column (clamped norm: no gradient
    through the norm)."""
    g = torch.Generator().manual_seed(2)
    B, C, n2, n1, eps = 3, 256, 128, 64, 1e-8
    s0 = torch.randn(B, C, n2, generator=g).to(dev)
    t0 = torch.randn(B, n1, C, generator=g).to(dev).transpose(1, 2)               # a (B,C,n1) view of point-major rows
    s0[:, :, 5] = 0
    w = torch.randn(B, n2, n1, generator=g).to(dev)

    def run(fn):
        s, t = s0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
        cos = fn(s, t)
        (cos * w).sum().backward()
        return cos.detach(), s.grad, t.grad

    def stock(s, t):
        tn = t / t.norm(dim=1, keepdim=True).clamp_min(eps)
        sn = s / s.norm(dim=1, keepdim=True).clamp_min(eps)
        return torch.bmm(sn.transpose(1, 2), tn)

    ref = run(stock)
    got = run(lambda s, t: train_ops._CosMap.apply(s, t, eps))
    keep = [j for j in range(n2) if j != 5]
    for a, b, name in zip(got, ref, ("cos", "ds", "dt")):
        assert a.shape == b.shape
        if name == "ds":                            # the zero column's gradient is G t / eps (1e8 times the others): compared apart
            za, zb = a[:, :, 5], b[:, :, 5]
            assert float((za - zb).abs().max()) <= 1e-5 * float(zb.abs().max()), "ds of the clamped column"
            a, b = a[:, :, keep], b[:, :, keep]
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-7, name


@pytest.mark.parametrize("B,N,E,hot", [(48, 1024, 16384, False), (3, 512, 8192, True), (2, 100, 37, False), (1, 16, 16384, True),
                                       (2, 2048, 4096, False), (2, 3000, 5000, False), (4, 128, 2048, True), (1, 1, 700, False),
                                       (2, 1024, 65536, True)])          # more entries than the bitonic network holds: counting sort only
def test_scatter_csr_is_the_stable_sort_by_bin(dev, B, N, E, hot):
    """ptt_scatter_csr_i32 (counting sort up to 2048 bins, bitonic above): order = the entries sorted by (bin, entry) — exactly
    numpy's stable argsort — and start = the bins' first slots; hot: most entries in a few bins (ball-query padding repeats a
    group's first index)."""
    rs = np.random.RandomState(E + N)
    idx = rs.randint(0, N, size=(B, E))
    if hot:
        idx[:, ::2] = idx[:, :1] % max(1, N // 8)
    order, start = ops.scatter_csr(torch.from_numpy(idx.astype(np.int32)).to(dev), N)
    order, start = order.cpu().numpy(), start.cpu().numpy()
    for b in range(B):
        assert np.array_equal(order[b], np.argsort(idx[b], kind="stable"))
        assert np.array_equal(start[b], np.searchsorted(np.sort(idx[b]), np.arange(N + 1), side="left"))


def test_clip_adam_update_is_seen_by_the_weight_pack_cache_and_by_autograd(dev):
    """The fused update writes the parameters through raw pointers: their version counters must advance as for an in-place torch
    op, or train_ops.packed (keyed on the version) would keep multiplying by the previous step's weights."""
    W = torch.nn.Parameter(torch.randn(128, 256, device=dev))
    o = ClipAdam([W], lr=0.1)
    for _ in range(2):
        before = train_ops.packed(W).clone()
        v0 = W._version
        W.grad = torch.randn_like(W)
        o.step(max_norm=10.0)
        assert W._version > v0
        after = train_ops.packed(W)
        assert torch.equal(after, ops.pack_weight(W.detach())) and not torch.equal(after, before)


def test_three_training_steps_with_clip_adam_track_the_stock_optimiser(dev):
    """The full tracker for three steps: ClipAdam against clip_grad_norm_ + torch.optim.Adam. The twin that the stock optimiser drives
    takes over the ClipAdam model's weights before every forward pass, so both see bit-identical losses and gradients (the row
    kernels are deterministic) and only the two optimisers' arithmetic is compared: every parameter after every step within 1e-6
    (a thousandth of the learning rate). Free-running twins are not comparable beyond two steps: a 1e-7 difference flips a near-tie
    of the box head's furthest point sampling in one of them (scripts/probes/three_step_diag.py). The losses of steps 2 and 3
    depend on the updated weights (packed for the MFMA kernels once per step)."""
    from ptt_amd.config import StubDataset, ptt_model_cfg
    from ptt_amd.models import build_network
    from ptt_amd.train_step import synthetic_train_batch
    kw = dict(lr=1e-3, betas=(0.5, 0.999), eps=1e-6)
    models, opts = [], []
    for fused in (True, False):
        torch.manual_seed(11)
        model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
        models.append(model)
        opts.append(ClipAdam(model.parameters(), **kw) if fused else torch.optim.Adam(model.parameters(), **kw))
    losses, worst = [], 0.0
    for step in range(3):
        models[1].load_state_dict(models[0].state_dict())
        run = []
        for model, opt in zip(models, opts):
            ret, _, _ = model(dict(synthetic_train_batch(20 + step, 4, dev)))
            opt.zero_grad(set_to_none=True)
            ret['loss'].backward()
            run.append(float(ret['loss'].detach()))
        assert run[0] == run[1], (step, run)
        for (n, p), (_, q) in zip(models[0].named_parameters(), models[1].named_parameters()):
            assert (p.grad is None) == (q.grad is None) and (p.grad is None or torch.equal(p.grad, q.grad)), n
        opts[0].step(max_norm=10.0)
        torch.nn.utils.clip_grad_norm_(models[1].parameters(), 10.0)
        opts[1].step()
        for (n, p), (_, q) in zip(models[0].named_parameters(), models[1].named_parameters()):
            d = float((p.detach() - q.detach()).abs().max())
            worst = max(worst, d)
            assert d <= 1e-6, (step, n, d)
        losses.append(run[0])
    print("ClipAdam vs clip_grad_norm_ + Adam over three steps: losses %s, largest parameter difference %.2e" % (losses, worst))
    assert abs(losses[1] - losses[0]) > 1e-3 * abs(losses[0]) and abs(losses[2] - losses[1]) > 1e-4 * abs(losses[1])   # new weights were seen


@pytest.mark.parametrize("R,C,want_dz", [(3 * 37 * 5, 64, True), (48 * 64 * 16, 256, False), (70001, 132, True), (64, 4, True)])
def test_sa_layer0_bn_backward_folded_into_the_weight_gradient_pass(dev, R, C, want_dz):
    """ptt_sa_z0_bnbwd_f32 against the two launches it replaces (ptt_bn_bwd_from_partials_f32, then ptt_linear_wgrad_f32 over dz0 and
    the relative coordinates): dz0, dgamma, dbeta bit-identical (the same arithmetic), d_wx to float32 summation-order accuracy; row
    counts that are no multiple of the row block, the chunk or 4 x the row groups."""
    g = torch.Generator().manual_seed(R + C)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    G, z, rel = r(R, C), r(R, C), r(R, 3)
    mean, invstd, gamma, a, b = r(C), r(C).abs() + 0.5, r(C), r(C), r(C)
    part = torch.randn(17, 2, C, generator=g, dtype=torch.float64).to(dev)
    dz_ref, dgamma_ref, dbeta_ref = ops.bn_bwd_from_partials(part, G, z, mean, invstd, gamma, a, b)
    dwx_ref = dz_ref.double().t() @ rel.double()
    G2 = G.clone()
    dz, dwx, dgamma, dbeta = ops.sa_z0_bnbwd(part, G2, z, rel, mean, invstd, gamma, a, b, want_dz)
    assert torch.equal(dgamma, dgamma_ref) and torch.equal(dbeta, dbeta_ref)
    if want_dz:
        assert dz.data_ptr() == G2.data_ptr() and torch.equal(dz, dz_ref)            # written over the gradient
    else:
        assert dz is None and torch.equal(G2, G)                                      # nothing written
    assert float((dwx.double() - dwx_ref).abs().max()) <= 1e-5 * float(dwx_ref.abs().max()) + 1e-6 * np.sqrt(R)
    again = ops.sa_z0_bnbwd(part, G.clone(), z, rel, mean, invstd, gamma, a, b, want_dz)
    assert torch.equal(again[1], dwx)                                                 # fixed summation order


@pytest.mark.parametrize("B,n2,n1,C", [(2, 7, 5, 12), (3, 128, 64, 256), (1, 4, 1, 4), (2, 33, 10, 64)])
def test_xcorr_layer0_bn_backward_folded_into_its_consumer_is_bit_identical(dev, B, n2, n1, C):
    """ptt_xcorr_z0_bnbwd_f32 (z0 recomputed, BatchNorm backward applied on the fly) against ptt_bn_bwd_from_partials_f32 followed by
    ptt_xcorr_z0_bwd_f32 on the stored z0: every output bit-identical, for search counts that are no multiple of the four rows in
    flight."""
    g = torch.Generator().manual_seed(B * 1000 + n2)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    P, cos, w = r(B, n1, C), r(B, n2, n1).clamp(-1, 1), r(C)
    z0 = ops.xcorr_z0(P, cos, w)
    G = r(B * n2 * n1, C)
    mean, invstd, gamma, a, b = r(C), r(C).abs() + 0.5, r(C), r(C), r(C)
    part = torch.randn(9, 2, C, generator=g, dtype=torch.float64).to(dev)
    dz_ref, dgamma_ref, dbeta_ref = ops.bn_bwd_from_partials(part, G, z0, mean, invstd, gamma, a, b)
    ref = ops.xcorr_z0_bwd(dz_ref, cos, w, B, n2, n1)
    got = ops.xcorr_z0_bnbwd(part, G, P, cos, w, mean, invstd, gamma, a, b)
    for x, y, name in zip(got, tuple(ref) + (dgamma_ref, dbeta_ref), ("dP", "dcos", "dw", "dgamma", "dbeta")):
        assert torch.equal(x, y), name
