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
