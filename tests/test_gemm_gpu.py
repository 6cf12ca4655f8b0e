"""The training step's row GEMMs (ptt_amd/csrc/gemm_ops.hip, through the C ABI) against float64 on the GPU:
ptt_rows_gemm_f32 — values, the fused column statistics (incl. a channel whose mean is 300 standard deviations: the
float32 E[y^2] - mean^2 form keeps no digit there), the deferred BatchNorm + ReLU input, the bias / ReLU / residual and the
ReLU-mask epilogues, ragged row counts (rows past the end neither stored nor counted), bit-reproducibility;
ptt_linear_wgrad2_f32 — values with and without the input transform, accumulation, reproducibility.
Shapes: every instantiation (128-channel chunks with one and two column tiles per wave, 64-channel chunks with 128- and
64-wide outputs), column groups (N = 512, 1536), K chunks (K = 256, 512), one partial tile, fewer rows than one tile."""
import numpy as np
import pytest
import torch

from ptt_amd import ops

pytestmark = pytest.mark.gpu

SHAPES = [(4096, 128, 256), (70001, 128, 128), (33333, 256, 256), (20000, 512, 512), (50011, 64, 64), (40000, 64, 128),
          (9999, 256, 512), (6144, 256, 1536), (130, 128, 256), (64, 64, 64), (5000, 192, 320)]


@pytest.mark.parametrize("R,K,C", SHAPES)
def test_rows_gemm_values_statistics_and_epilogues(dev, R, K, C):
    g = torch.Generator(device="cpu").manual_seed(R + K + C)
    x = torch.randn(R, K, generator=g).to(dev)
    w = (torch.randn(C, K, generator=g) / K ** 0.5).to(dev)
    wp = ops.pack_weight(w)
    assert ops.rows_gemm_supported(R, K, C)
    ref = x.double() @ w.double().t()
    y, st = ops.rows_gemm(x, wp, C, want_stats=True)
    assert float((y.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())              # measured <= 1.1e-6
    mean, var, invstd = ops.bn_finish_partials(st, R, 1e-5)
    v64, m64 = torch.var_mean(y.double(), 0, unbiased=False)
    assert float((mean.double() - m64).abs().max()) <= 1e-6
    assert float(((var.double() - v64) / v64).abs().max()) <= 3e-6                               # measured <= 1e-6
    y2, st2 = ops.rows_gemm(x, wp, C, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(st, st2)                                           # fixed summation order
    # a nearly constant output channel set: |mean| = 300 standard deviations
    xb = torch.cat([x[:, :K - 1] * 1e-2, torch.ones(R, 1, device=dev)], 1).contiguous()
    wb = w.clone()
    wb[:, K - 1] = 3.0
    yb, stb = ops.rows_gemm(xb, ops.pack_weight(wb), C, want_stats=True)
    _, vb, _ = ops.bn_finish_partials(stb, R, 1e-5)
    v64b = torch.var(yb.double(), 0, unbiased=False)
    assert float(((vb.double() - v64b) / v64b).abs().max()) <= 3e-6                              # measured <= 4e-7
    # deferred activation on the input + bias + ReLU + residual in the launch
    a = (torch.rand(K, generator=g) + 0.5).to(dev)
    b = (torch.randn(K, generator=g) * 0.3).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    res = torch.randn(R, C, generator=g).to(dev)
    ref3 = torch.relu(torch.relu(x.double() * a.double() + b.double()) @ w.double().t() + bias.double()) + res.double()
    y3 = ops.rows_gemm(x, wp, C, in_scale=a, in_shift=b, bias=bias, relu=True, residual=res)
    assert float((y3.double() - ref3).abs().max()) <= 2e-6 * float(ref3.abs().max())
    # with a deferred activation the rows past the end are not zeros inside the kernel: the statistics must not see them
    y4, st4 = ops.rows_gemm(x, wp, C, in_scale=a, in_shift=b, want_stats=True)
    m4, v4, _ = ops.bn_finish_partials(st4, R, 1e-5)
    v64d, m64d = torch.var_mean(y4.double(), 0, unbiased=False)
    assert float((m4.double() - m64d).abs().max()) <= 1e-6 and float(((v4.double() - v64d) / v64d).abs().max()) <= 3e-6
    # the ReLU-backward epilogue: out = mask > 0 ? x W^T : 0 and its column sums
    mask = torch.randn(R, C, generator=g).to(dev)
    out, colsum = ops.rows_gemm_masked(x, wp, C, mask, want_colsum=True)
    refm = torch.where(mask.double() > 0, ref, torch.zeros_like(ref))
    assert float((out.double() - refm).abs().max()) <= 3e-6 * float(ref.abs().max())
    assert float((colsum.double() - refm.sum(0)).abs().max()) <= 2e-5 * float(refm.abs().sum(0).max())


@pytest.mark.parametrize("R,Cout,Cin", [(393216, 256, 256), (98304, 512, 512), (200000, 256, 128), (50000, 128, 256), (33333, 128, 128),
                                         (6144, 512, 512), (2500, 256, 256)])
def test_wgrad2_values_transform_accumulate_reproducible(dev, R, Cout, Cin):
    g = torch.Generator(device="cpu").manual_seed(R)
    dz = torch.randn(R, Cout, generator=g).to(dev)
    x = torch.randn(R, Cin, generator=g).to(dev)
    got = ops.linear_wgrad(dz, x)
    ref = dz.double().t() @ x.double()
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6 * np.sqrt(R)     # measured 6e-7
    assert torch.equal(got, ops.linear_wgrad(dz, x))
    a = (torch.rand(Cin, generator=g) + 0.5).to(dev)
    b = (torch.randn(Cin, generator=g) * 0.3).to(dev)
    got2 = ops.linear_wgrad(dz, x, x_scale=a, x_shift=b)
    ref2 = dz.double().t() @ torch.relu(x.double() * a.double() + b.double())
    assert float((got2.double() - ref2).abs().max()) <= 2e-6 * float(ref2.abs().max()) + 1e-6 * np.sqrt(R)
    acc = ops.linear_wgrad(dz, x, out=got.clone(), accumulate=True)
    torch.testing.assert_close(acc, 2 * got, rtol=1e-6, atol=1e-6)


def test_bn_statistics_launch_also_writes_the_activation_constants_and_the_running_statistics(dev):
    """ops.bn_stats / bn_finish_partials with `bn`: the launch that forms the batch statistics also writes a = gamma * invstd,
    b = beta - mean * a (bit-equal to the element-wise expressions) and does nn.BatchNorm's training bookkeeping — compared
    with the separate launches (ops.bn_update_running) and with torch.nn.BatchNorm1d itself."""
    import copy
    torch.manual_seed(5)
    for R, C in ((4096, 128), (777, 64), (50000, 256)):
        x = torch.randn(R, C, device=dev) * 1.7 + 0.4
        bn = torch.nn.BatchNorm1d(C, momentum=0.1).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
            bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
        ref, sep = copy.deepcopy(bn), copy.deepcopy(bn)
        mean0, var0, inv0 = ops.bn_stats(x, bn.eps)
        ops.bn_update_running(sep, mean0, var0, torch.full((1,), float(R), dtype=torch.float64, device=dev))
        v_before = bn.running_mean._version
        mean, var, inv, a, b = ops.bn_stats(x, bn.eps, bn=bn)
        assert torch.equal(mean, mean0) and torch.equal(var, var0) and torch.equal(inv, inv0)
        assert torch.equal(a, bn.weight.detach() * inv0) and torch.equal(b, bn.bias.detach() - mean0 * a)
        assert bn.running_mean._version > v_before                         # eval-mode caches key on the version
        assert int(bn.num_batches_tracked) == 1 == int(sep.num_batches_tracked)
        torch.testing.assert_close(bn.running_mean, sep.running_mean, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(bn.running_var, sep.running_var, rtol=1e-6, atol=1e-7)
        ref(x)
        torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
        if ops.rows_gemm_supported(R, 64, C, 64, C):                        # the same tail behind the GEMM's partial sums
            w = torch.randn(C, 64, device=dev) * 0.1
            xin = torch.randn(R, 64, device=dev)
            z, part = ops.rows_gemm(xin, ops.pack_weight(w), C, want_stats=True)
            bn2 = copy.deepcopy(ref)
            m0, v0, i0 = ops.bn_finish_partials(part, R, bn2.eps)
            m1, v1, i1, a1, b1 = ops.bn_finish_partials(part, R, bn2.eps, bn=bn2)
            assert torch.equal(m0, m1) and torch.equal(v0, v1) and torch.equal(i0, i1)
            assert torch.equal(a1, bn2.weight.detach() * i0) and torch.equal(b1, bn2.bias.detach() - m0 * a1)
            assert int(bn2.num_batches_tracked) == 2
            n = float(R)
            want = 0.9 * ref.running_var + 0.1 * v0 * (n / (n - 1))
            torch.testing.assert_close(bn2.running_var, want, rtol=1e-6, atol=1e-7)


def test_pack_plan_repacks_every_registered_weight_in_one_launch(dev):
    """train_ops.packed: the first call per (view, transposed?) packs alone and registers; after an in-place update (version
    bump) ONE ptt_pack_weights_f32 launch re-packs the whole registered set — bit-equal to ops.pack_weight of each view,
    for plain, transposed, reshaped-4-D and column-sliced weights; a replaced parameter storage rebuilds the job table."""
    from ptt_amd import train_ops
    torch.manual_seed(6)
    plan = train_ops._pack_plans.setdefault(dev, train_ops._PackPlan())
    conv = torch.nn.Parameter(torch.randn(128, 67, 1, 1, device=dev))
    lin = torch.nn.Parameter(torch.randn(512, 256, device=dev))
    small = torch.nn.Parameter(torch.randn(64, 3, device=dev))

    def views():
        w0 = conv.reshape(128, -1)
        return [(w0, False), (w0, True), (w0[:, 3:], False), (w0[:, 3:], True), (w0[:, 0:3], False), (lin, False), (lin, True),
                (small, False), (small, True)]

    def expect(W, tr):
        w = W.detach()
        return ops.pack_weight((w.t() if tr else w).contiguous())

    launches = {"n": 0}
    real = ops.pack_weights

    def counting(*a):
        launches["n"] += 1
        return real(*a)

    ops.pack_weights = counting
    try:
        for W, tr in views():                                            # first sight: packed one by one
            assert torch.equal(train_ops.packed(W, tr), expect(W, tr))
        assert launches["n"] == 0
        for step in range(2):
            with torch.no_grad():
                for p in (conv, lin, small):
                    p.add_(torch.randn_like(p) * 0.1)                   # the optimiser's in-place update
            got = [train_ops.packed(W, tr) for W, tr in views()]
            assert launches["n"] == step + 1, launches                   # one launch for the whole set, cache hits after it
            for (W, tr), g in zip(views(), got):
                assert torch.equal(g, expect(W, tr)), (tuple(W.shape), tr)
        assert train_ops.packed(lin, False) is train_ops.packed(lin, False)
        with torch.no_grad():
            lin.data = lin.data.clone()                                   # new storage under the same parameter object
            lin.add_(1.0)
        assert torch.equal(train_ops.packed(lin, True), expect(lin, True))
        assert torch.equal(train_ops.packed(conv.reshape(128, -1), True), expect(conv.reshape(128, -1), True))
    finally:
        ops.pack_weights = real
    assert len(plan.entries) >= 9


@pytest.mark.parametrize("rows,K,N,ns", [(4096, 128, 256, 32), (70016, 256, 256, 16), (8192, 256, 256, 64), (49152 + 64, 64, 128, 32),
                                         (3 * 64 * 7, 128, 256, 64)])
def test_pool_epilogue_equals_the_pooling_pass(dev, rows, K, N, ns):
    """ptt_rows_gemm_pool_f32 + ptt_pool_select_f32 (the max-pool from the extrema the GEMM's epilogue takes) against
    ptt_rows_gemm_f32 + ptt_pool_rows_f32 (a pooling pass over z): the same z and statistics bit for bit, the same pooled values
    bit for bit (relu(a z + b) is monotone in z), and an arg-max that holds the pooled value — for positive, negative and zero
    BatchNorm scales, with duplicated rows (exact ties) in the input."""
    g = torch.Generator().manual_seed(rows + K + ns)
    x = torch.randn(rows, K, generator=g)
    x[5::7] = x[3::7][:x[5::7].shape[0]]                      # duplicated rows: ties inside groups
    x = x.to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    ia, ib = (torch.rand(K, generator=g) + 0.5).to(dev), (torch.randn(K, generator=g) * 0.1).to(dev)
    a = torch.randn(N, generator=g).to(dev)
    a[::5] = 0.0
    b = (torch.randn(N, generator=g) * 0.3).to(dev)
    wp = ops.pack_weight(w)
    z0, st0 = ops.rows_gemm(x, wp, N, in_scale=ia, in_shift=ib, want_stats=True)
    assert ops.rows_gemm_pool_supported(rows, K, N, x.stride(0), ns, x=x)
    z1, st1, ext = ops.rows_gemm_pool(x, wp, N, ia, ib, ns)
    assert torch.equal(z0, z1) and torch.equal(st0, st1)
    p0, arg0 = ops.pool_rows(z0, ns, a, b)
    p1, arg1 = ops.pool_select(ext, a, b)
    assert torch.equal(p0, p1)
    act = torch.relu(z1 * a + b).view(rows // ns, ns, N)
    picked = act.gather(1, arg1.long().unsqueeze(1)).squeeze(1)
    assert torch.equal(picked, act.max(dim=1)[0])            # the arg-max holds the maximum
    pos = (a > 0)
    first = (z1.view(rows // ns, ns, N) == ext[0].unsqueeze(1)).float().argmax(dim=1)
    assert torch.equal(arg1[:, pos], first[:, pos].int())    # and is the FIRST row of the extremum
