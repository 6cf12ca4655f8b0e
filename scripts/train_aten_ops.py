"""Dev tool: which ATen operators (not the library's own kernels) does one training step of the full tracker still launch —
counts and device time per operator, with the Python call site of the most frequent ones."""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
trainer = DataParallelTrainer(model, dev)
batch = synthetic_train_batch(100, 48, dev)
for _ in range(3):
    trainer.step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.step(batch)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0]
by_op = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    by_op[e.key][0] += e.count; by_op[e.key][1] += e.device_time_total
print("ATen operators with device time in ONE step (count, device us):")
for k, (c, t) in sorted(by_op.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-32s %5d  %9.1f us" % (k, c, t))
print("\ncall sites of the heaviest (operator, stack) pairs:")
for e in sorted(ev, key=lambda e: -e.device_time_total)[:28]:
    site = [s for s in e.stack if "/ptt_amd/" in s or "bench.py" in s or "train_step" in s]
    print("  %-24s x%-4d %8.1f us  %s" % (e.key, e.count, e.device_time_total, (site[0] if site else (e.stack[0] if e.stack else "?"))[-110:]))
