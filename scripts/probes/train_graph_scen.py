"""Dev probe: the captured step against the eager one in separate processes. Scenarios: schedule same|rotate x trainer eager|graph|graph+twin."""
import os, subprocess, sys
if len(sys.argv) < 2:
    out = {}
    for sched in ("same", "rotate"):
        for mode in ("eager", "graph", "twin"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), mode, sched], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("LOSSES")]
            out[(sched, mode)] = line[0] if line else "rc=%d %s" % (r.returncode, (r.stdout + r.stderr).strip().splitlines()[-2:])
            print(sched, mode, out[(sched, mode)], flush=True)
        print(sched, "graph==eager:", out[(sched, "graph")] == out[(sched, "eager")], " twin==eager:", out[(sched, "twin")] == out[(sched, "eager")], flush=True)
    sys.exit(0)
mode, sched = sys.argv[1], sys.argv[2]
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "8"))


def make(graph):
    torch.manual_seed(1)
    model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
    return DataParallelTrainer(model, dev, graph=graph)


tr = make(mode != "eager")
twin = make(False) if mode == "twin" else None
batches = [synthetic_train_batch(100 + k, B, dev) for k in range(3)]
losses = []
for k in range(8):
    b = batches[k % 3 if sched == "rotate" else 0]
    if twin is not None:
        twin.step(b)
    losses.append(float(tr.step(b).detach()))
    torch.cuda.synchronize()
print("LOSSES", " ".join("%.7f" % l for l in losses))
print("captured" if tr.captured is not None else "not captured")
