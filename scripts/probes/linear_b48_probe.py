"""Dev probe (round 4): the short linear launches of the 48-frame step on linear_small_kernel / rows_gemm_kernel (what ops.linear picks)
against rowjobs_kernel with 1, 2 or 4 column tiles per workgroup — microseconds per launch inside a hipGraph chain."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops
dev = torch.device("cuda:0")


def graph_time(fn, n_chain=30, reps=100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_chain):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n_chain * 1e6


for name, rows, K, C in (("qkv", 6144, 256, 1536), ("qkv64", 3072, 256, 1536), ("fc2", 6144, 512, 256), ("fc2_64", 3072, 512, 256),
                         ("cov_s", 6144, 256, 256), ("cov_t", 3072, 256, 256), ("box_hoist", 6144, 257, 256), ("hoist2_s", 12288, 256, 128),
                         ("hoist2_t", 6144, 256, 128), ("hoist1_s", 24576, 128, 128), ("hoist1_t", 12288, 128, 128)):
    ld = (K + 3) // 4 * 4
    x = torch.randn(rows, ld, device=dev)[:, :K]
    w = torch.randn(C, K, device=dev) / K ** 0.5
    b = torch.zeros(C, device=dev)
    wp = ops.pack_weight(w)
    o = torch.empty((rows, C), device=dev)
    saved = ops.ROW_JOB_MAX_ROWS
    line = "%-10s %6d x %4d -> %4d  ops.linear %6.2f us" % (name, rows, K, C, graph_time(lambda: ops.linear(x, wp, C, None, b, False, None, out=o)))
    for cw in (1, 2, 4):
        t = graph_time(lambda: ops.row_jobs([ops.row_job(wp, C, x=x, shift=b, out=o, col_tiles=cw)]))
        line += "   rowjobs cw=%d %6.2f" % (cw, t)
    fl = 2.0 * rows * K * C
    print(line + "   (%.1f GFLOP)" % (fl / 1e9), flush=True)

print("the same shapes on the persistent row GEMM (ptt_rows_gemm_f32) where it takes them:")
for name, rows, K, C in (("qkv", 6144, 256, 1536), ("qkv64", 3072, 256, 1536), ("fc2", 6144, 512, 256), ("fc2_64", 3072, 512, 256),
                         ("cov_s", 6144, 256, 256), ("hoist2_s", 12288, 256, 128), ("hoist2_t", 6144, 256, 128), ("hoist1_t", 12288, 128, 128)):
    x = torch.randn(rows, K, device=dev)
    wp = ops.pack_weight(torch.randn(C, K, device=dev) / K ** 0.5)
    b = torch.zeros(C, device=dev)
    o = torch.empty((rows, C), device=dev)
    if ops.rows_gemm_supported(rows, K, C, K, C, x=x):
        print("%-10s %6d x %4d -> %4d  rows_gemm %6.2f us" % (name, rows, K, C, graph_time(lambda: ops.rows_gemm(x, wp, C, bias=b, out=o))), flush=True)
