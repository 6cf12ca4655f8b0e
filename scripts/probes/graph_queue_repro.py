#!/usr/bin/env python
"""Pure-torch reproduction attempt of the hipGraphLaunch crash (no ptt_amd code): one captured graph with BRANCHES parallel
branches (fork from the capture stream to side streams, trivial kernels, join), instantiated and replayed, after PRE throw-away
graphs of the same shape were created (kept alive or dropped). The suspect: a graph whose internal parallel streams land on the
SAME hardware queue — forced with GPU_MAX_HW_QUEUES=1/2 in the environment.

    GPU_MAX_HW_QUEUES=2 python scripts/probes/graph_queue_repro.py [BRANCHES=3] [PRE=0] [keep|drop] [null|side]"""
import faulthandler
import sys

import torch

faulthandler.enable()


import os

KIND = os.environ.get("REPRO_KIND", "torch")      # torch: element-wise torch ops only; ptt: the side branches run a kernel of
#                                                    libptt_hip.so (furthest point sampling); long: 64 torch ops per branch;
#                                                    copy: a device-to-device copy_ node at the join


def side_work(x, k, pts):
    if KIND == "ptt":
        from ptt_amd import ops
        return ops.furthest_point_sampling(pts, 64).float().sum() + x[:1]
    y = x * (k + 2)
    for _ in range(64 if KIND == "long" else 4):
        y = y + 1
    return y


def build(branches, dev):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    pts = torch.rand(4, 1024, 3, device=dev) + 0.1
    x = torch.ones(1 << 20, device=dev)
    static = torch.zeros(1 << 20, device=dev)
    sides = [torch.cuda.Stream(device=dev) for _ in range(branches - 1)]
    g = torch.cuda.CUDAGraph()
    warm = torch.cuda.Stream(device=dev)
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        (x + 1).sum()
        side_work(x, 0, pts)
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        outs = []
        for k, s in enumerate(sides):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(side_work(x, k, pts))
        z = x - 1
        for _ in range(4):
            z = z * 1.5
        for s in sides:
            main.wait_stream(s)
        tot = z
        for y in outs:
            tot = tot + y
        if KIND == "copy":
            static.copy_(tot)
    return g, tot, sides


def main():
    branches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    keep = (sys.argv[3] if len(sys.argv) > 3 else "drop") == "keep"
    on_null = (sys.argv[4] if len(sys.argv) > 4 else "null") == "null"
    dev = torch.device("cuda:0")
    kept = []
    for k in range(pre):
        g, tot, sides = build(branches, dev)
        g.replay()
        torch.cuda.synchronize()
        if keep:
            kept.append((g, tot, sides))
        del g, tot, sides
    print("[repro] %d graphs before, %s" % (pre, "kept" if keep else "dropped"), flush=True)
    g, tot, sides = build(branches, dev)
    st = torch.cuda.current_stream() if on_null else torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for _ in range(20):
            g.replay()
    torch.cuda.synchronize()
    print("[repro] branches %d: replayed, sum %.1f: PASSED" % (branches, float(tot.sum())), flush=True)


if __name__ == "__main__":
    main()
