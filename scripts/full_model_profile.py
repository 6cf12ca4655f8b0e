#!/usr/bin/env python
"""Dev: per-kernel time of one eager full-tracker forward at B=48 (torch profiler, device time)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.hot_path import randomize_
from ptt_amd.models import build_network
dev = torch.device("cuda:0"); B = 48
trk = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), 0).to(dev).eval()
s, t = synth.frames(0, B, 2048, 1024); s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
f = lambda: trk({'search_points': s, 'template_points': t, 'batch_size': B})
with torch.no_grad():
    for _ in range(3): f()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5): f()
        torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / 5.0, e.count // 5) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
tot = sum(r[1] for r in rows)
print("total device us/step: %.0f" % tot)
for k, us, n in rows[:28]:
    print("%9.1f us  x%-3d %s" % (us, n, k[:90]))
