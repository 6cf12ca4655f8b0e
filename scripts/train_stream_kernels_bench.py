"""Dev: the training step's streaming kernels of round 4 at the step's own shapes, a few launches each — the command
scripts/pmc_passes.sh profiles for them (FETCH_SIZE / WRITE_SIZE: does each tensor cross the fabric once?).
    bash scripts/pmc_passes.sh gpurun_out/x_pmc - "python scripts/train_stream_kernels_bench.py"
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops
dev = torch.device("cuda:0")
B, it = 48, 4
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
def timed(name, fn, mb):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); us = (time.perf_counter() - t) / it * 1e6
    print("%-44s %8.1f us  %6.0f MB algorithmic  %5.2f TB/s" % (name, us, mb, mb / us))
# SA0's narrow weight gradients (786432 rows)
for cout in (64, 128):
    dz, x = r(786432, cout), r(786432, 64)
    a, b = r(64), r(64)
    timed("wgrad_stream  786432 x %3d^T x 64" % cout, lambda: ops.linear_wgrad(dz, x, x_scale=a, x_shift=b), (dz.numel() + x.numel()) * 4 / 1e6)
# layer-0 BatchNorm backward folded into its consumer
for name, R, C, want in (("sa_z0_bnbwd  SA0 786432 x 64 (no dz0)", 786432, 64, False), ("sa_z0_bnbwd  SA1 393216 x 128 (dz0 out)", 393216, 128, True)):
    G, z, rel = r(R, C), r(R, C), r(R, 3)
    m, s, ga, aa, bb = r(C), r(C).abs() + 0.5, r(C), r(C), r(C)
    part = torch.randn(64, 2, C, generator=g, dtype=torch.float64).to(dev)
    timed(name, lambda: ops.sa_z0_bnbwd(part, G, z, rel, m, s, ga, aa, bb, want), ((2 + int(want)) * R * C + 3 * R) * 4 / 1e6)
n2, n1, C = 128, 64, 256
P, cos, w = r(B, n1, C), r(B, n2, n1).clamp(-1, 1), r(C)
G = r(B * n2 * n1, C)
m, s, ga, aa, bb = r(C), r(C).abs() + 0.5, r(C), r(C), r(C)
part = torch.randn(64, 2, C, generator=g, dtype=torch.float64).to(dev)
timed("xcorr_z0_bnbwd  393216 x 256 (z0 recomputed)", lambda: ops.xcorr_z0_bnbwd(part, G, P, cos, w, m, s, ga, aa, bb), G.numel() * 4 / 1e6)
idx = torch.randint(0, 512, (B, 256 * 32), generator=g).to(torch.int32).to(dev)
timed("scatter_csr  48 x 8192 entries into 512 bins", lambda: ops.scatter_csr(idx, 512), idx.numel() * 12 / 1e6)
