"""Probe: two pipelined graphs round-robin — equal stream priorities vs one way at high priority."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import synth
from ptt_amd.hot_path import FrameHotPath, PipelinedHotPath, kitti_model_cfg, randomize_
dev = torch.device("cuda:0")
model = randomize_(FrameHotPath(kitti_model_cfg()), seed=0).to(dev).eval()
s, t = synth.frames(1000, 48, 2048, 1024)
s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
def bench(fn, n=600):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for prios in ((0, 0), (-1, 0), (-1, -1)):
    st = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    pipes = []
    for k in range(2):
        with torch.cuda.stream(st[k]):
            pipes.append(PipelinedHotPath(model, s, t))
    torch.cuda.synchronize()
    i = [0]
    def alt():
        k = i[0] % 2; i[0] += 1
        with torch.cuda.stream(st[k]):
            pipes[k]()
    print("stream priorities %s: %.4f ms/step" % (prios, bench(alt)))
    del pipes
