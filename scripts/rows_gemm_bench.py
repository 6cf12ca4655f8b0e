"""Dev tool: the row GEMMs of the training step at their real shapes — ptt_rows_gemm_f32 (persistent, software-pipelined;
forward / input gradient), ptt_linear_f32 (the inference kernel it replaces there), ptt_linear_wgrad2_f32 / ptt_linear_wgrad_f32
(weight gradient) against torch.mm (hipBLASLt fp32) on the same device: TFLOP/s per shape, plus a float64 check of every
new kernel's result (values, fused statistics, deferred-activation input, bias / ReLU / residual epilogue, ragged row counts)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops
dev = torch.device("cuda:0")


PMC = "--pmc" in sys.argv            # under rocprofv3 --pmc: a few launches of the three big shapes only


def timeit(fn, iters=10):
    if PMC:
        fn(); fn(); torch.cuda.synchronize(); return 1.0
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def check():
    torch.manual_seed(0)
    worst = 0.0
    for R, K, C in [(4096, 128, 256), (70001, 128, 128), (33333, 256, 256), (20000, 512, 512), (50011, 64, 64), (40000, 64, 128),
                    (9999, 256, 512), (6144, 256, 1536), (130, 128, 256), (64, 64, 64), (5000, 192, 320)]:
        x = torch.randn(R, K, device=dev)
        w = torch.randn(C, K, device=dev) / K ** 0.5
        wp = ops.pack_weight(w)
        if not ops.rows_gemm_supported(R, K, C):
            print("unsupported", R, K, C); continue
        ref = x.double() @ w.double().t()
        y, st = ops.rows_gemm(x, wp, C, want_stats=True)
        e = float((y.double() - ref).abs().max() / ref.abs().max())
        mean, var, invstd = ops.bn_finish_partials(st, R, 1e-5)
        v64, m64 = torch.var_mean(ref, 0, unbiased=False)
        e2 = float((mean.double() - m64).abs().max()), float(((var.double() - v64) / v64).abs().max())
        # a nearly constant output channel (|mean| = 300 std): E[y^2] - mean^2 in float32 would keep no digit of the variance
        xb = torch.cat([x[:, :K - 1] * 1e-2, torch.ones(R, 1, device=dev)], 1).contiguous()
        wb = w.clone(); wb[:, K - 1] = 3.0
        yb, stb = ops.rows_gemm(xb, ops.pack_weight(wb), C, want_stats=True)
        mb, vb, _ = ops.bn_finish_partials(stb, R, 1e-5)
        v64b, m64b = torch.var_mean(yb.double(), 0, unbiased=False)
        e2 = e2 + (float(((vb.double() - v64b) / v64b).abs().max()),)
        y2, st2 = ops.rows_gemm(x, wp, C, want_stats=True)
        same = torch.equal(y, y2) and torch.equal(st, st2)
        # deferred activation on the input + bias + relu + residual
        a = torch.rand(K, device=dev) + 0.5; b = torch.randn(K, device=dev) * 0.3
        bias = torch.randn(C, device=dev); res = torch.randn(R, C, device=dev)
        ref3 = torch.relu(torch.relu(x.double() * a.double() + b.double()) @ w.double().t() + bias.double()) + res.double()
        y3 = ops.rows_gemm(x, wp, C, in_scale=a, in_shift=b, bias=bias, relu=True, residual=res)
        e3 = float((y3.double() - ref3).abs().max() / ref3.abs().max())
        dz = torch.randn(R, C, device=dev)
        g = ops.linear_wgrad(dz, x)
        gref = dz.double().t() @ x.double()
        e4 = float((g.double() - gref).abs().max() / gref.abs().max())
        g2 = ops.linear_wgrad(dz, x, x_scale=a, x_shift=b)
        gref2 = dz.double().t() @ torch.relu(x.double() * a.double() + b.double())
        e5 = float((g2.double() - gref2).abs().max() / gref2.abs().max())
        ok = e < 2e-6 and e3 < 2e-6 and e4 < 1e-5 and e5 < 1e-5 and e2[0] < 1e-5 and e2[1] < 1e-4 and e2[2] < 1e-4 and same
        worst = max(worst, e, e3)
        print("check R=%d K=%d N=%d: y %.1e  act/bias/relu/res %.1e  mean %.1e var %.1e (offset channel %.1e)  wgrad %.1e / %.1e  reproducible %s  %s"
              % (R, K, C, e, e3, e2[0], e2[1], e2[2], e4, e5, same, "ok" if ok else "FAIL"))
    return worst


def bench():
    shapes = [(786432, 64, 64), (786432, 64, 128), (393216, 128, 128), (393216, 128, 256), (393216, 256, 256), (196608, 256, 256),
              (98304, 512, 512), (49152, 512, 512), (98304, 256, 1536), (49152, 256, 256), (6144, 256, 256)]
    if PMC:
        shapes = [(393216, 128, 256), (393216, 256, 256), (98304, 512, 512)]
    for R, K, C in shapes:
        x = torch.randn(R, K, device=dev); w = torch.randn(C, K, device=dev) / K ** 0.5
        wp = ops.pack_weight(w); wt = w.t().contiguous()
        dz = torch.randn(R, C, device=dev)
        gf = 2.0 * R * K * C / 1e9
        line = "R=%d K=%d N=%d:" % (R, K, C)
        ms = timeit(lambda: ops.linear(x, wp, C)); line += "  linear %.3f ms %.0f TF" % (ms, gf / ms)
        if ops.rows_gemm_supported(R, K, C):
            out = torch.empty(R, C, device=dev)
            ms = timeit(lambda: ops.rows_gemm(x, wp, C, out=out)); line += " | rows_gemm %.3f ms %.0f TF" % (ms, gf / ms)
            ms = timeit(lambda: ops.rows_gemm(x, wp, C, want_stats=True, out=out)); line += " (+stats %.3f ms %.0f TF)" % (ms, gf / ms)
        out = torch.empty(R, C, device=dev)
        ms = timeit(lambda: torch.mm(x, wt, out=out)); line += " | torch.mm %.3f ms %.0f TF" % (ms, gf / ms)
        ops.WGRAD2 = False
        ms = timeit(lambda: ops.linear_wgrad(dz, x)); line += " || wgrad(r2) %.3f ms %.0f TF" % (ms, gf / ms)
        ops.WGRAD2 = True
        ms = timeit(lambda: ops.linear_wgrad(dz, x)); line += " | wgrad2 %.3f ms %.0f TF" % (ms, gf / ms)
        dzt = dz.t()
        ms = timeit(lambda: torch.mm(dzt, x)); line += " | torch.mm(dz^T, x) %.3f ms %.0f TF" % (ms, gf / ms)
        print(line, flush=True)


if __name__ == "__main__":
    if "--no-check" not in sys.argv:
        check()
    if "--no-bench" not in sys.argv:
        bench()
