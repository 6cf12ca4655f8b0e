"""The captured training step against the eager one: two trainers built from one seed step through the same batches; parameters,
moments, BatchNorm buffers, loss and clipped norm must be bit-identical after every step. Then the host time to queue a step.
    python scripts/train_graph_check.py [B] [steps]            PTT_CHECK_COLLECTIVE=1: one-rank RCCL group, two graphs + all-reduce"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
collective = os.environ.get("PTT_CHECK_COLLECTIVE", "0") == "1"
if collective:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1)


def make(graph):
    torch.manual_seed(1)
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    return DataParallelTrainer(model, dev, graph=graph, force_ddp=collective)


eager, graphed = make(False), make(True)
batches = [synthetic_train_batch(100 + k, B, dev) for k in range(3)]
bad = 0
for k in range(steps):
    b = batches[k % 3]
    le = eager.step(b).clone()
    lg = graphed.step(b).clone()
    torch.cuda.synchronize()
    same = torch.equal(le, lg)
    for (n, p), q in zip(eager.tracker.state_dict().items(), graphed.tracker.state_dict().values()):
        if not torch.equal(p, q):
            same = False
            print("  step %d: %s differs (max %.3e)" % (k, n, (p.double() - q.double()).abs().max().item()))
            break
    same = same and torch.equal(eager.sink.flat, graphed.sink.flat) and torch.equal(eager.optimizer.last_norm, graphed.optimizer.last_norm)
    print("step %d: %s loss %.6f norm %.4f identical=%s" % (k, "graph" if graphed.captured is not None else "eager", float(lg), float(graphed.optimizer.last_norm), same), flush=True)
    bad += 0 if same else 1
print("captured:", graphed.captured is not None, "graph steps:", graphed.graph_steps, "mismatching steps:", bad)

for name, tr in (("eager", eager), ("graph", graphed)):
    b = batches[0]
    for _ in range(3):
        tr.step(b)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s B=%d: host issue %.3f ms per step, device done %.3f ms per step" % (name, B, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3), flush=True)
sys.exit(1 if bad or graphed.captured is None else 0)
