#!/usr/bin/env python
"""Per-frame latency at B=1 (the reference's sequential tracking loop runs the model one frame at a time,
tools/eval_utils/eval_tracking_utils.py:231-264): eager vs hipGraph replay, hot path and full tracker."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.hot_path import FrameHotPath, GraphedHotPath, kitti_model_cfg, randomize_
from ptt_amd.models import build_network

dev = torch.device("cuda:0")


def bench(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for NS, NT in ((1024, 512), (2048, 1024)):
    s, t = synth.frames(0, 1, NS, NT)
    s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
    hot = randomize_(FrameHotPath(kitti_model_cfg()), 0).to(dev).eval()
    trk = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), 0).to(dev).eval()
    full = lambda a, b: trk({'search_points': a, 'template_points': b, 'batch_size': 1})
    with torch.no_grad():
        e_hot = bench(lambda: hot(s, t)); e_full = bench(lambda: full(s, t))
    g_hot = GraphedHotPath(hot, s, t); g_full = GraphedHotPath(full, s, t)
    print("B=1 %d+%d pts: hot path eager %.3f ms, graph %.3f ms | full tracker eager %.3f ms, graph %.3f ms (%.0f frames/s)"
          % (NS, NT, e_hot, bench(g_hot), e_full, bench(g_full), 1e3 / bench(g_full)))
