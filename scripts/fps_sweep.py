"""Dev tool (needs a -DPTT_DEV build): FPS iteration time vs threads per cloud at the tracking-loop sizes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops, synth
dev = torch.device("cuda:0")
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for B in (1, 48):
    for N, m in ((1024, 512), (512, 256), (2048, 512)):
        s, _ = synth.frames(4, B, N, 64, K_s=max(64, int(N * 0.3)))
        xyz = torch.from_numpy(s).to(dev)
        ref = None
        for T in (0, 64, 128, 256, 512, 1024):
            if T: os.environ["PTT_FPS_T"] = str(T)
            else: os.environ.pop("PTT_FPS_T", None)
            out = ops.furthest_point_sampling(xyz, m)
            if ref is None: ref = out
            assert torch.equal(out, ref)
            ms = timeit(lambda: ops.furthest_point_sampling(xyz, m))
            print("B=%d N=%d m=%d T=%s: %.4f ms  %.3f us/iter" % (B, N, m, T or "default", ms, ms * 1e3 / (m - 1)))
