"""Dev probe: tracking throughput with G lockstep groups of 48 tracklets advancing alternately (run_overlapped) against
one group after the other (TrackletRunner.run) — same tracklets, same results."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptt_amd import synth
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.hot_path import randomize_
from ptt_amd.models import build_network
from ptt_amd.tracklet_runner import TrackletRunner, run_overlapped
dev = torch.device("cuda:0")
tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=0).to(dev).eval()
B, T = 48, 30
for G in (1, 2, 3, 4):
    tr = [synth.tracklet(9000 + k, T) for k in range(B * G)]
    runners = [TrackletRunner(tracker, dev, batch=B) for _ in range(G)]
    warm = [(c[:4], b[:4]) for c, b in tr]
    run_overlapped(runners, warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); res_o = run_overlapped(runners, tr); torch.cuda.synchronize(); dt_o = time.perf_counter() - t0
    runners[0].run(warm[:B]); torch.cuda.synchronize()
    t0 = time.perf_counter(); res_s = runners[0].run(tr); torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
    same = all(np.array_equal(a[-1][0], b[-1][0]) for a, b in zip(res_o, res_s))
    frames = B * G * (T - 1)
    print("G=%d (%3d tracklets): alternating groups %.0f frames/s (%.3f ms per 48-step), one group after the other %.0f frames/s; "
          "same final boxes: %s" % (G, B * G, frames / dt_o, dt_o / (G * (T - 1)) * 1e3, frames / dt_s, same))
