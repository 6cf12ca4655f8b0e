"""Dev tool (needs a -DPTT_DEV build: PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force): where the time of ONE iteration of
the farthest-point-sampling chain goes, by ABLATION: the kernel is re-instantiated with one link of the iteration's dependency
chain removed (PTT_FPS_ABL, see fps_kernel's header; the indices are then wrong, the time per iteration is what is read) and timed
on one cloud — the one-tracklet shapes: search 1024 -> 512 (256 threads x 4 points: 4 waves, a barrier per iteration) and template
512 -> 256 (64 threads x 8 points: one wave). The ABL = 0 instantiation is the production kernel, instruction for instruction."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops, synth
dev = torch.device("cuda:0")
LINKS = [(0, "the kernel as shipped"), (32, "scan over ONE pair of points instead of all P"), (1, "no wave maximum (DPP chain)"), (2, "no ballot + find-first"),
         (3, "neither"), (16, "no barrier (slots still written and read)"), (4, "no exchange between the waves (slot write, barrier, fold)"),
         (8, "no read-back of the winner's coordinates from LDS"), (12, "no exchange, no read-back"), (7, "no wave maximum, ballot, exchange"),
         (15, "none of the four"), (47, "none of the four, scan of one pair: the loop's skeleton")]


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for N, m in ((1024, 512), (512, 256)):
    s, _ = synth.frames(4, 1, N, 64, K_s=max(64, int(N * 0.6)))
    xyz = torch.from_numpy(s).to(dev)
    base = None
    for abl, what in LINKS:
        if abl:
            os.environ["PTT_FPS_ABL"] = str(abl)
        else:
            os.environ.pop("PTT_FPS_ABL", None)
        ms = min(timeit(lambda: ops.furthest_point_sampling(xyz, m)) for _ in range(3))
        it = ms * 1e3 / (m - 1)
        base = it if base is None else base
        print("FPS %4d -> %3d  ABL %2d  %.4f ms  %.3f us per iteration  (%+.3f us, %+5.1f %%)  %s" % (N, m, abl, ms, it, it - base, 100 * (it - base) / base, what), flush=True)
os.environ.pop("PTT_FPS_ABL", None)
