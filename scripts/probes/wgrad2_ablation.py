"""Dev probe (needs PTT_GEMM_FLAGS=-DPTT_GEMM_DEV): where wgrad2_kernel<4,2>'s time goes — the launch with parts compiled out
(PTT_WG2_EXP bits: 1 no staging writes, 2 no row requests, 4 no barriers, 8 fragments of step 0 only, 16 no partial stores;
wrong results), on operands the Infinity Cache holds (a small R) and on operands from HBM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for R, Cin, Cout in [(98304, 512, 512), (393216, 256, 256)]:
    x = torch.randn(R, Cin, device=dev); dz = torch.randn(R, Cout, device=dev)
    for exp in ("0", "1", "2", "3", "4", "7", "8", "15", "16", "31"):
        if exp == "0": os.environ.pop("PTT_WG2_EXP", None)
        else: os.environ["PTT_WG2_EXP"] = exp
        ms = timeit(lambda: ops.linear_wgrad_partials(dz, x))
        print("R=%d %dx%d EXP=%-2s %.3f ms  %.0f TF" % (R, Cout, Cin, exp, ms, 2.0 * R * Cin * Cout / 1e9 / ms), flush=True)
