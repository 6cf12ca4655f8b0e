"""Probe: two PipelinedHotPath graphs replayed alternately on two streams (two independent batches in flight) vs one."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import synth
from ptt_amd.hot_path import FrameHotPath, PipelinedHotPath, kitti_model_cfg, randomize_
dev = torch.device("cuda:0")
model = randomize_(FrameHotPath(kitti_model_cfg()), seed=0).to(dev).eval()
s, t = synth.frames(1000, 48, 2048, 1024)
s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
def bench(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
p1 = PipelinedHotPath(model, s, t)
print("one pipeline: %.4f ms/step" % bench(lambda: p1()))
for W in (2, 3, 4):
    st = [torch.cuda.Stream(device=dev) for _ in range(W)]
    pipes = []
    for k in range(W):
        with torch.cuda.stream(st[k]):
            pipes.append(PipelinedHotPath(model, s, t))
    torch.cuda.synchronize()
    i = [0]
    def alt():
        k = i[0] % W; i[0] += 1
        with torch.cuda.stream(st[k]):
            pipes[k]()
    print("%d pipelines alternating on %d streams: %.4f ms/step" % (W, W, bench(alt, 600)))
    del pipes
