"""Dev: a Conv1d stack as one launch (ptt_rows_mlp_f32) vs one ptt_linear_f32 launch per layer, at one frame and at 48."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for rows in (128, 6144):
    for widths in ([259, 256, 256, 259], [256, 256, 256, 1], [256, 256, 256]):
        x = torch.randn(rows, widths[0], device=dev)
        layers = []
        for i, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
            w = torch.randn(cout, cin, device=dev) / cin ** 0.5
            layers.append((ops.pack_weight(w), None, torch.randn(cout, device=dev), cin, cout, i < len(widths) - 2))
        def chain():
            y = x
            for wp, sc, sh, cin, cout, relu in layers:
                y = ops.linear(y, wp, cout, sc, sh, relu)
            return y
        one = lambda: ops.rows_mlp(x, layers)
        print("rows %5d %-22s per-layer launches %7.1f us   one launch %7.1f us" % (rows, widths, timeit(chain), timeit(one)))
