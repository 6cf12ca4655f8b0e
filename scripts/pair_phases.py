#!/usr/bin/env python
"""Dev tool: per-phase cycle stamps of pt_attn_pair_kernel (PTT_DEBUG_STAMPS hook)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import ops, synth
from tests.util import transformer_params
dev = torch.device("cuda:0"); B, N = 48, 128
P = {k: v.to(dev).contiguous() for k, v in transformer_params(1).items()}
s, _ = synth.frames(1, B, N, 64, K_s=N); xyz = torch.from_numpy(s).to(dev)
knn = ops.knn(xyz, 16); qkv = torch.randn(B, N, 1536, device=dev)
packs = [ops.pack_weight(P[k]) for k in ("fc_delta.2.weight", "fc_gamma.0.weight", "fc_gamma.2.weight")]
wd1p = ops.pack_delta0(P["fc_delta.0.weight"], P["fc_delta.0.bias"])
fn = lambda: ops.pt_attn_pair(xyz, knn, qkv, wd1p, packs[0], P["fc_delta.2.bias"], packs[1], P["fc_gamma.0.bias"], packs[2], P["fc_gamma.2.bias"], 512, False)
for _ in range(3): fn()
if len(sys.argv) > 1: os.environ["PTT_PAIR_LDS_PAD"] = sys.argv[1]
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["PTT_DEBUG_STAMPS"] = "%x" % buf.data_ptr()
fn(); torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(4096, 8)[:3072]
d = np.diff(st, axis=1)
names = ["P0 knn+hdelta", "G1 delta gemm", "E1 bar+t", "G2 gamma0 gemm", "E2 g write", "G3 gamma2 gemm", "E3 softmax"]
print("cycles per phase (median / p10 / p90 over %d workgroups); total median %d" % (len(st), np.median(st[:, 7] - st[:, 0])))
for i, n in enumerate(names):
    print("  %-16s %8.0f %8.0f %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
t0 = st[:, 0].min()
order = np.argsort(st[:, 0])
print("start offsets of first 12 WGs (cycles):", (st[order[:12], 0] - t0).tolist())
print("kernel span cycles:", st[:, 7].max() - t0)
import time
def bench(tag):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10; print(tag, "%.4f ms  %.1f TF" % (ms, 2.0*B*N*16*(3*512+3*512*512)/ms/1e9))
os.environ.pop("PTT_DEBUG_STAMPS")
for sg in (0, 4, 8, 11, 16, 24):
    os.environ["PTT_PAIR_STAGGER"] = str(sg); bench("stagger=%d" % sg)
