"""Dev probe (round 4): what ONE launch of a dependent chain costs inside a hipGraph on this box, and what the short
GEMM launches of a one-frame chain cost on linear_small_kernel against rowjobs_kernel.
    python scripts/probes/launch_floor_probe.py
"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops

dev = torch.device("cuda:0")


def graph_time(fn, n_chain, reps=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_chain):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_chain * 1e6


x1 = torch.zeros(64, device=dev)
print("chain of trivial torch kernels (x.add_(1) on 64 floats): %.2f us per node" % graph_time(lambda: x1.add_(1.0), 100))

rs = np.random.RandomState(0)
for rows, K, C in ((128, 256, 256), (128, 512, 512), (128, 256, 1536), (64, 512, 512), (2048, 512, 512), (1024, 512, 512), (512, 128, 128)):
    x = torch.from_numpy(rs.standard_normal((rows, K)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rs.standard_normal((C, K)) / np.sqrt(K)).astype(np.float32)).to(dev)
    b = torch.zeros(C, device=dev)
    wp = ops.pack_weight(w)
    # a dependent chain: ping-pong between two buffers of the same shape when K == C, else the same launch repeated
    o = torch.empty((rows, C), device=dev)
    t_lin = graph_time(lambda: ops.linear(x, wp, C, None, b, True, None, out=o), 50)
    line = "%5d x %4d -> %4d   linear %.2f us" % (rows, K, C, t_lin)
    for cw in (0, 1, 2, 4):
        t = graph_time(lambda: ops.row_jobs([ops.row_job(wp, C, x=x, shift=b, act=1, out=o, col_tiles=cw)]), 50)
        line += "   rowjobs cw=%d %.2f us" % (cw, t)
    print(line, flush=True)

# the three pair-row launches of a transformer block at N = 128 / 64
for N in (128, 64):
    D, P = 512, N
    qkv = torch.randn((P, 3 * D), device=dev)
    knn = torch.stack([torch.randperm(N)[:16] for _ in range(P)]).to(torch.int32).to(dev)
    pos = torch.randn((P * 16, D), device=dev)
    rel = torch.randn((P * 16, 3), device=dev)
    w1 = torch.randn((D, 4), device=dev)
    wp = ops.pack_weight(torch.randn((D, D), device=dev) / 22.6)
    b = torch.zeros(D, device=dev)
    g = torch.empty((P * 16, D), device=dev)
    res = torch.empty((P, D), device=dev)
    t1 = graph_time(lambda: ops.row_jobs([ops.row_job(wp, D, prologue=1, rel=rel, w1=w1, K=D, shift=b, out=g)]), 30)
    t2 = graph_time(lambda: ops.row_jobs([ops.row_job(wp, D, prologue=2, qkv=qkv, knn=knn, pos=pos, k_off=D, N=N, K=D, shift=b, act=1, out=g)]), 30)
    t3 = graph_time(lambda: ops.row_jobs([ops.row_job(wp, D, x=g, epilogue=1, qkv=qkv, knn=knn, pos=pos, v_off=2 * D, N=N, sm_scale=0.0442, out=res)]), 30)
    print("N = %3d: delta job %.2f us, gamma0 (pair input) job %.2f us, gamma2 (softmax / aggregate) job %.2f us" % (N, t1, t2, t3), flush=True)
