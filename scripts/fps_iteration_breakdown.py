"""Dev tool (needs a -DPTT_DEV build: PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force): where the cycles of ONE iteration of
the farthest-point-sampling chain go. fps_kernel stamps the shader-cycle counter at its segment boundaries for iterations 256..263 of
cloud 0 (ptt_dev_fps_stamps); this prints the mean cycles per segment for the one-tracklet search cloud (1024 -> 512: 256 threads x 4
points, 4 waves, one barrier per iteration) and the template cloud (512 -> 256 would stop before iteration 256, so 512 -> 384 is run:
64 threads x 8 points, one wave, no barrier), and the measured time per iteration beside them."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import _lib, ops, synth
dev = torch.device("cuda:0")
lib = _lib.lib()
if not hasattr(lib, "ptt_dev_fps_stamps"):
    sys.exit("needs a build with PTT_HIP_FLAGS=-DPTT_DEV")
SEG_W = ["scan of the thread's points (packed fp32 sub/mul/add, min, select)", "wave maximum (fused DPP)", "ballot + find-first", "winner's slot -> LDS",
         "workgroup barrier", "read + fold the 4 wave slots", "winner's coordinates back from LDS", "loop back (index to LDS, branch)"]
SEG_1 = ["scan of the thread's points (packed fp32 sub/mul/add, min, select)", "wave maximum (fused DPP)", "ballot + find-first",
         "readlane of the winner's index", None, None, "winner's coordinates back from LDS", "loop back (index to LDS, branch)"]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for N, m, names in ((1024, 512, SEG_W), (512, 384, SEG_1)):
    s, _ = synth.frames(4, 1, N, 64, K_s=max(64, int(N * 0.6)))
    xyz = torch.from_numpy(s).to(dev)
    ops.furthest_point_sampling(xyz, m)
    torch.cuda.synchronize()
    buf = (ctypes.c_int * 64)()
    assert lib.ptt_dev_fps_stamps(buf) == 0
    st = np.array(list(buf), np.int64).reshape(8, 8)
    ms = timeit(lambda: ops.furthest_point_sampling(xyz, m))
    us_iter = ms * 1e3 / (m - 1)
    order = [k for k in range(8) if names[k if k < 7 else 7] is not None or k == 7]
    used = [0, 1, 2, 3, 4, 5, 6, 7] if names is SEG_W else [0, 1, 2, 3, 4, 7]
    d = np.diff(st[:, used], axis=1) % (1 << 20)                       # 20-bit counter
    whole = (np.diff(st[:, 0]) % (1 << 20)).astype(float)              # top of iteration j to top of iteration j + 1
    seg = d.mean(0)
    back = whole.mean() - seg.sum()
    labels = [names[k] for k in used[1:]]
    if names is SEG_W:
        labels = [SEG_W[0], SEG_W[1], SEG_W[2], SEG_W[3], SEG_W[4], SEG_W[5], SEG_W[6]]
    else:
        labels = [SEG_1[0], SEG_1[1], SEG_1[2], SEG_1[3], SEG_1[6]]
    print("FPS %d -> %d, one cloud: %.4f ms = %.3f us per iteration; one iteration = %.0f shader cycles (stamped iterations 256..263; "
          "%.2f GHz implied)" % (N, m, ms, us_iter, whole.mean(), whole.mean() / us_iter / 1e3))
    for lab, c in zip(labels, seg):
        print("    %-75s %6.0f cycles  %4.1f %%" % (lab, c, 100 * c / whole.mean()))
    print("    %-75s %6.0f cycles  %4.1f %%" % ("loop back: index to LDS, stamps' own cost, branch", back, 100 * back / whole.mean()))
