"""Dev probe: is the training step waiting for the host? Twenty steps queued without a synchronisation: host time per step to
ISSUE them against the time until the device has finished them."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
tr = DataParallelTrainer(model, dev)
B = int(os.environ.get("PROBE_B", "48"))
batch = synthetic_train_batch(100, B, dev)
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host issue %.2f ms per step, device done %.2f ms per step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3), flush=True)
