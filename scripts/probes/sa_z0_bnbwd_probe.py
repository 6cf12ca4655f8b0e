"""Dev probe: ptt_sa_z0_bnbwd_f32 on the training step's six SA shapes — time per launch and GB/s of the bytes it must move
(G and z0 once each + rel), (+ the row scatter of a level with point features) against the launches it replaces (bn_bwd_from_partials apply + scatter_rows_det
+ linear_wgrad). Note: uniform random neighbour indices — the step's ball-query indices are heavily unbalanced (padding)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops
dev = torch.device("cuda:0")
B = 48
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, N, M, ns, C, has_term in (("SA0_s", 1024, 512, 32, 64, False), ("SA1_s", 512, 256, 32, 128, True), ("SA2_s", 256, 128, 32, 128, True),
                                    ("SA0_t", 512, 256, 32, 64, False), ("SA1_t", 256, 128, 32, 128, True), ("SA2_t", 128, 64, 32, 128, True)):
    R = B * M * ns
    g = torch.randn(R, C, device=dev); z = torch.randn(R, C, device=dev); rel = torch.randn(R, 3, device=dev)
    idx = torch.randint(0, N, (B, M, ns), device=dev, dtype=torch.int32)
    mean, invstd, gamma, a, b = (torch.randn(C, device=dev) for _ in range(5))
    part = torch.randn(64, 2, C, device=dev, dtype=torch.float64)
    csr = ops.scatter_csr(idx.view(B, M * ns), N) if has_term else None
    def new():
        dz, dwx, dg, db = ops.sa_z0_bnbwd(part, g, z, rel, mean, invstd, gamma, a, b, has_term)
        if has_term: ops.scatter_rows_det(dz.view(B, M * ns, C), idx.view(B, M * ns), N, csr)
    t_new = bench(new)
    def old():
        dz, dg, db = ops.bn_bwd_from_partials(part, g, z, mean, invstd, gamma, a, b)
        if has_term: ops.scatter_rows_det(dz.view(B, M * ns, C), idx.view(B, M * ns), N, csr)
        ops.linear_wgrad(dz, rel)
    t_old = bench(old)
    mb = (2 * R * C + 3 * R) * 4 / 1e6
    print("%-6s rows %7d C %3d  fused %7.1f us (%5.2f TB/s)   apply + scatter + wgrad %7.1f us" % (name, R, C, t_new, mb / t_new, t_old))
