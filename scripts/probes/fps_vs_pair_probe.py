"""Probe: how much a concurrent FPS launch (48 clouds, one workgroup each) slows the pair kernel on another stream."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import ops, synth
from tests.util import transformer_params
dev = torch.device("cuda:0"); B, N = 48, 128
P = {k: v.to(dev).contiguous() for k, v in transformer_params(1).items()}
s, _ = synth.frames(1, B, N, 64, K_s=N); xyz = torch.from_numpy(s).to(dev)
knn = ops.knn(xyz, 16); qkv = torch.randn(B, N, 1536, device=dev)
packs = [ops.pack_weight(P[k]) for k in ("fc_delta.2.weight", "fc_gamma.0.weight", "fc_gamma.2.weight")]
wd1p = ops.pack_delta0(P["fc_delta.0.weight"], P["fc_delta.0.bias"])
pair = lambda: ops.pt_attn_pair(xyz, knn, qkv, wd1p, packs[0], P["fc_delta.2.bias"], packs[1], P["fc_gamma.0.bias"], packs[2], P["fc_gamma.2.bias"], 512, False)
c, _ = synth.frames(4, B, 2048, 64, K_s=600); cloud = torch.from_numpy(c).to(dev)
side = torch.cuda.Stream()
def run(n_pair, fps_every):
    for _ in range(3): pair()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n_pair):
        if fps_every and i % fps_every == 0:
            with torch.cuda.stream(side):
                ops.furthest_point_sampling(cloud, 512)
        pair()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_pair * 1e3
print("pair kernel alone:                         %.4f ms" % run(60, 0))
print("one FPS (0.28 ms alone) per 3 pair launches: %.4f ms per pair launch" % run(60, 3))
print("one FPS per pair launch:                   %.4f ms per pair launch" % run(60, 1))
