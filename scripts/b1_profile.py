import os, sys, torch
sys.path.insert(0, '/root/repo')
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.hot_path import randomize_
from ptt_amd.models import build_network
dev = torch.device("cuda:0"); B = 1
trk = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), 0).to(dev).eval()
s, t = synth.frames(0, B, 1024, 512); s, t = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
f = lambda: trk({'search_points': s, 'template_points': t, 'batch_size': B})
with torch.no_grad():
    for _ in range(3): f()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): f()
        torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / 10.0, e.count // 10) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
print("total device us/frame: %.0f  launches %d" % (sum(r[1] for r in rows), sum(r[2] for r in rows)))
for k, us, n in rows[:22]:
    print("%8.1f us  x%-3d %s" % (us, n, k[:100]))
