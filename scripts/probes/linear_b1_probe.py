"""Dev: per-launch time of the small linear launches of one tracklet frame, inside a hipGraph of 20 dependent launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops
dev = torch.device("cuda:0")
for rows, K, C in ((128, 256, 256), (128, 256, 1536), (64, 512, 256), (512, 128, 128), (6144, 256, 256)):
    ws = [ops.pack_weight(torch.randn(C if i % 2 == 0 else K, K if i % 2 == 0 else C, device=dev) / 16) for i in range(20)]
    x = torch.randn(rows, K, device=dev)
    def chain():
        y = x
        for i, w in enumerate(ws):
            y = ops.linear(y, w, C if i % 2 == 0 else K)
        return y
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): chain()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = chain()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print("rows %5d  %4d <-> %4d : %6.2f us per launch in a graph of 20" % (rows, K, C, (time.perf_counter() - t0) / 50 / 20 * 1e6))
