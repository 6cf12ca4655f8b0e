"""Probe: what the side-stream FPS of the next batch costs the step — the same two-way pipelined replay with the sampling
stage replaced by a copy of cached indices (wrong for new inputs: timing only)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import synth
from ptt_amd.hot_path import FrameHotPath, InterleavedHotPath, kitti_model_cfg, randomize_
dev = torch.device("cuda:0")
model = randomize_(FrameHotPath(kitti_model_cfg()), seed=0).to(dev).eval()
s, t = synth.frames(1000, 48, 2048, 1024)
s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
def bench(fn, n=600):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
p = InterleavedHotPath(model, s, t, ways=2)
print("two ways, FPS of the next batch on the side stream: %.4f ms/step" % bench(lambda: p()))
del p
class NoFps(object):
    def __init__(self, m):
        self.m = m
        with torch.no_grad():
            self.cached = [x.clone() for x in m.sample(s, t)]
    def sample(self, a, b):
        return [x.clone() for x in self.cached]
    def __call__(self, a, b, inds=None):
        return self.m(a, b, inds)
p = InterleavedHotPath(NoFps(model), s, t, ways=2)
print("two ways, sampling stage = a copy of cached indices:   %.4f ms/step" % bench(lambda: p()))
