"""Dev probe: the sequence of tests/test_train_graph_gpu.py::test_other_batch_shapes..., with / without a synchronisation after every step."""
import os, subprocess, sys
if len(sys.argv) < 2:
    for mode in ("sync", "nosync", "nosync_notwin"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True)
        print(mode, "rc", r.returncode, "|", " ".join((r.stdout + r.stderr).strip().splitlines()[-6:])[-700:], flush=True)
    sys.exit(0)
mode = sys.argv[1]
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
dev = torch.device("cuda:0")


def make(graph):
    torch.manual_seed(1)
    return DataParallelTrainer(build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train(), dev, graph=graph)


g = make(True)
e = make(False) if "notwin" not in mode else None
b8, b4 = synthetic_train_batch(100, 8, dev), synthetic_train_batch(101, 4, dev)
seq = [b8] * 5 + [b4, b8, b8]
for k, b in enumerate(seq):
    if e is not None:
        e.step(b)
    l = g.step(b)
    if mode == "sync":
        torch.cuda.synchronize()
        print("step", k, "ok", float(l.detach()), "graph_steps", g.graph_steps, flush=True)
torch.cuda.synchronize()
print("done; graph_steps", g.graph_steps, "loss", float(l.detach()))
if e is not None:
    print("params equal", all(torch.equal(p, q) for p, q in zip(e.tracker.state_dict().values(), g.tracker.state_dict().values())))
