"""Dev probe: the training step with and without the template branch's level-0 sampling on a side stream (same process, same box,
alternating blocks of steps)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
trainer = DataParallelTrainer(model, dev)
batch = synthetic_train_batch(100, 48, dev)
for _ in range(5): trainer.step(batch)
res = {True: [], False: []}
for rep in range(6):
    for flag in (True, False):
        model.backbone_3d.overlap_branches = flag
        for _ in range(2): trainer.step(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): trainer.step(batch)
        torch.cuda.synchronize(); res[flag].append((time.perf_counter() - t) / 20 * 1e3)
for flag in (True, False):
    v = sorted(res[flag]); print("template FPS on a side stream = %-5s  ms/step median %.3f  (min %.3f max %.3f)" % (flag, v[len(v) // 2], v[0], v[-1]))
