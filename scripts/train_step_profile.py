#!/usr/bin/env python
"""Dev: per-kernel device time of one training step (fwd + bwd + Adam) of the full tracker at B=48."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
dev = torch.device("cuda:0"); B = int(os.environ.get("B", 48))
if os.environ.get("NO_MIOPEN"): torch.backends.cudnn.enabled = False
torch.manual_seed(1)
model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=1e-6)
s, t = synth.frames(100, B, 1024, 512, K_s=200, K_t=100)
def step():
    batch = {'search_points': torch.from_numpy(s).to(dev), 'template_points': torch.from_numpy(t).to(dev), 'batch_size': B,
             'cls_label': (torch.rand(B, 1024, device=dev) > 0.7).float(), 'reg_label': torch.randn(B, 4, device=dev) * 0.3}
    ret, _, _ = model(batch)
    loss = ret['loss'].mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("ms/step %.1f" % ((time.perf_counter() - t0) / 5 * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / 2.0, e.count // 2) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[1])
print("total device us/step: %.0f" % sum(r[1] for r in rows))
for k, us, n in rows[:40]:
    print("%10.1f us  x%-4d %s" % (us, n, k[:110]))
