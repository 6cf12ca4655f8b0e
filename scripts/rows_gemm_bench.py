"""Dev tool (tile sweep needs a -DPTT_DEV build): the row GEMMs of the training step — ptt_linear_f32 (forward / input
gradient) and ptt_linear_wgrad_f32 at the shared-MLP shapes, TFLOP/s per shape and tile."""
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
shapes = [(786432, 64, 64), (786432, 64, 128), (393216, 128, 128), (393216, 128, 256), (393216, 256, 256), (98304, 512, 512)]
for R, K, C in shapes:
    x = torch.randn(R, K, device=dev); w = ops.pack_weight(torch.randn(C, K, device=dev) / K ** 0.5)
    dz = torch.randn(R, C, device=dev)
    line = "R=%d K=%d Cout=%d:" % (R, K, C)
    for tile in (None, "11", "12", "21", "22"):
        if tile: os.environ["PTT_LINEAR_TILE"] = tile
        else: os.environ.pop("PTT_LINEAR_TILE", None)
        ms = timeit(lambda: ops.linear(x, w, C))
        line += "  lin[%s] %.3f ms %.0f TF" % (tile or "def", ms, 2.0 * R * K * C / ms / 1e9)
    os.environ.pop("PTT_LINEAR_TILE", None)
    ms = timeit(lambda: ops.linear_wgrad(dz, x))
    line += "  | wgrad %.3f ms %.0f TF" % (ms, 2.0 * R * K * C / ms / 1e9)
    ms = timeit(lambda: torch.mm(x, torch.empty(K, C, device=dev)))
    line += "  | torch.mm %.3f ms %.0f TF" % (ms, 2.0 * R * K * C / ms / 1e9)
    print(line)
