"""Dev tool: device time of a training step per autograd / aten operator (torch.profiler)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        trainer.step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)))
for e in rows[:45]:
    t = getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))
    st = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    print("%-60s calls %5d  total %8.3f ms/step  self %8.3f ms/step" % (e.key[:60], e.count // 3, t / 3e3, st / 3e3))
