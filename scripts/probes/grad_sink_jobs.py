"""Dev probe: what the gradient sink of one training step (configs[3]) holds — contributions per size class and their partial bytes."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
from ptt_amd import train_ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
tr = DataParallelTrainer(model, dev)
batch = synthetic_train_batch(100, 48, dev)
seen = []
flush0 = train_ops.GradSink.flush
def flush(self):
    seen[:] = list(self.jobs)
    return flush0(self)
train_ops.GradSink.flush = flush
tr.step(batch)
agg = collections.Counter(); cnt = collections.Counter()
for dst, cols, ld, n, ptr, nch in seen:
    agg[(n, nch)] += n * nch * 4; cnt[(n, nch)] += 1
tot = sum(agg.values())
print("jobs %d, partial bytes %.1f MB" % (len(seen), tot / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
    print("n=%8d chunks=%5d x%3d  %8.1f MB" % (k[0], k[1], cnt[k], v / 1e6))
