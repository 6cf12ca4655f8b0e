"""Dev tool: the row-wise linear launches of the INFERENCE step at B = 48 (and one tracklet frame) on the three kernels —
linear_kernel (PTT_LINEAR_SMALL=0 in a -DPTT_DEV build), linear_small_kernel (<= 8192 rows) and the persistent
rows_gemm_kernel — microseconds per launch inside a hipGraph of 20 back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops
dev = torch.device("cuda:0")


def graph_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n)


shapes = [(24576, 128, 128), (12288, 128, 128), (12288, 256, 128), (6144, 256, 128), (6144, 256, 256), (3072, 256, 256),
          (6144, 256, 1536), (3072, 256, 1536), (6144, 512, 256), (3072, 512, 256), (3072, 256, 256),
          (2048, 512, 512), (1024, 512, 512), (512, 128, 128), (256, 256, 128), (128, 256, 256), (128, 256, 1536), (128, 512, 256)]
for R, K, C in shapes:
    x = torch.randn(R, K, device=dev); wp = ops.pack_weight(torch.randn(C, K, device=dev) / K ** 0.5)
    b = torch.randn(C, device=dev); out = torch.empty(R, C, device=dev)
    line = "R=%5d K=%3d N=%4d (%.2f GF):" % (R, K, C, 2e-9 * R * K * C)
    us = graph_us(lambda: ops.linear(x, wp, C, None, b, False, None, out=out)); line += "  linear(auto) %6.1f us %5.0f TF" % (us, 2e-6 * R * K * C / us)
    os.environ["PTT_LINEAR_SMALL"] = "0"
    us = graph_us(lambda: ops.linear(x, wp, C, None, b, False, None, out=out)); line += " | linear_kernel %6.1f us" % us
    os.environ.pop("PTT_LINEAR_SMALL")
    if ops.rows_gemm_supported(R, K, C):
        us = graph_us(lambda: ops.rows_gemm(x, wp, C, bias=b, out=out)); line += " | rows_gemm %6.1f us %5.0f TF" % (us, 2e-6 * R * K * C / us)
    print(line, flush=True)
