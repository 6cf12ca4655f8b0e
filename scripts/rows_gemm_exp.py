"""Dev tool (needs PTT_GEMM_FLAGS=-DPTT_GEMM_DEV python -m ptt_amd.build --force): timing-only ablations of rows_gemm_kernel
<1,2,2,128> (PTT_RG_EXP bits: 1 no stores, 2 no row fetch, 4 L1-resident weights, 8 no barriers) at the three big shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops
from rows_gemm_bench import timeit
dev = torch.device("cuda:0")
for R, K, C in [(393216, 128, 256), (393216, 256, 256), (98304, 512, 512)]:
    x = torch.randn(R, K, device=dev); wp = ops.pack_weight(torch.randn(C, K, device=dev) / K ** 0.5)
    out = torch.empty(R, C, device=dev)
    line = "R=%d K=%d N=%d:" % (R, K, C)
    for e in (0, 1, 2, 3, 4, 7, 8, 15):
        os.environ["PTT_RG_EXP"] = str(e)
        ms = timeit(lambda: ops.rows_gemm(x, wp, C, out=out))
        line += "  exp%d %.3f ms %.0f TF" % (e, ms, 2.0 * R * K * C / ms / 1e9)
    print(line, flush=True)
