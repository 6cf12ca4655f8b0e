"""Probe: with three batches in flight, does the template branch still need its own stream inside a batch?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import synth
from ptt_amd.hot_path import FrameHotPath, InterleavedHotPath, kitti_model_cfg, randomize_
dev = torch.device("cuda:0")
s, t = synth.frames(1000, 48, 2048, 1024)
s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
def bench(fn, n=600):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for overlap in (True, False):
        model = randomize_(FrameHotPath(kitti_model_cfg()), seed=0).to(dev).eval()
        model.overlap_branches = overlap
        for ways in (2, 3, 4):
            p = InterleavedHotPath(model, s, t, ways=ways)
            print("template branch on its own stream: %-5s ways %d: %.4f ms/step" % (overlap, ways, bench(lambda: p())))
            del p
