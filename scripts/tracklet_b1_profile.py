"""Dev tool: one tracklet through TrackletRunner at B = 1 — wall time per frame, host-side split, and (under
rocprofv3 --kernel-trace --stats) the device time per frame."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.hot_path import randomize_
from ptt_amd.models import build_network
from ptt_amd.tracklet_runner import TrackletRunner
dev = torch.device("cuda:0")
tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=0).to(dev).eval()
T = 200
tr = synth.tracklet(9000, T)
runner = TrackletRunner(tracker, dev, batch=1)
runner.run([(tr[0][:4], tr[1][:4])])
torch.cuda.synchronize()
loops = []
for _ in range(5):
    t0 = time.perf_counter(); runner.run([tr]); torch.cuda.synchronize(); loops.append((time.perf_counter() - t0) / (T - 1) * 1e3)
print("B=1 tracklet loop: median %.4f ms per frame over 5 passes of %d frames (%s)" % (sorted(loops)[2], T - 1, " ".join("%.4f" % v for v in loops)))
# the model graph alone, synchronised per frame
g = runner._frame if runner._frame is not None else runner._graph      # the whole-frame graph (crops + resampling + model + read-backs)
g = g.replay if hasattr(g, "replay") else g
t0 = time.perf_counter()
for _ in range(T):
    g(); torch.cuda.synchronize()
print("model graph replay + sync alone: %.4f ms per frame" % ((time.perf_counter() - t0) / T * 1e3))
t0 = time.perf_counter()
for _ in range(T):
    g()
torch.cuda.synchronize()
print("model graph back-to-back (no per-frame sync): %.4f ms per frame" % ((time.perf_counter() - t0) / T * 1e3))
