"""Dev tool: the weight-gradient launches of the training step at their real shapes — ops.linear_wgrad (ptt_linear_wgrad2_f32's
kernel + the fixed-order sum of its row-chunk partials), TFLOP/s per shape. PTT_WG2_FIRST (a -DPTT_GEMM_DEV build) skips the
larger output blocks: 0 = 256 x 256 per workgroup (default), 1 = 256 x 128, 2 = 128 x 256, 3 = 128 x 128."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for R, Cin, Cout in [(98304, 512, 512), (49152, 512, 512), (393216, 256, 256), (196608, 256, 256), (393216, 128, 256), (393216, 128, 128), (98304, 256, 512),
                     (6144, 256, 256)]:
    x = torch.randn(R, Cin, device=dev); dz = torch.randn(R, Cout, device=dev)
    ref = None
    line = "R=%d Cin=%d Cout=%d:" % (R, Cin, Cout)
    for first in os.environ.get("WG_FIRSTS", "0").split(","):
        os.environ["PTT_WG2_FIRST"] = first
        g = ops.linear_wgrad(dz, x)
        if ref is None:
            ref = dz.double().t() @ x.double()
        err = float((g.double() - ref).abs().max() / ref.abs().max())
        ms = timeit(lambda: ops.linear_wgrad(dz, x))
        line += "  first=%s %.3f ms %.0f TF (err %.1e)" % (first, ms, 2.0 * R * Cin * Cout / 1e9 / ms, err)
    print(line, flush=True)
