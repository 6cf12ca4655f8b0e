"""Dev tool: where a 48-tracklet lockstep step goes — model graph alone, device pre/post kernels, host."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd import synth, ops
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.hot_path import randomize_
from ptt_amd.models import build_network
from ptt_amd.tracklet_runner import TrackletRunner
dev = torch.device("cuda:0")
tracker = randomize_(build_network(ptt_model_cfg(), 1, StubDataset()), seed=0).to(dev).eval()
B, T = 48, 30
tr = [synth.tracklet(9000 + k, T) for k in range(B)]
runner = TrackletRunner(tracker, dev, batch=B)
runner.run([(c[:4], b[:4]) for c, b in tr])
torch.cuda.synchronize()
t0 = time.perf_counter(); runner.run(tr); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("full loop: %.3f ms per step" % (dt / (T - 1) * 1e3))
g = runner._graph
def tm(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("model graph replay + sync: %.3f ms" % tm(lambda: g()))
print("crop_compact (96 jobs) + sync: %.3f ms" % tm(lambda: ops.crop_compact(runner.crop_jobs_dev, 2 * B)))
print("regularize (96 jobs) + sync: %.3f ms" % tm(lambda: ops.regularize(runner.reg_jobs_dev, 2 * B, runner.draws)))
def copies():
    runner.result_host.copy_(g.out, non_blocking=True); runner.info_host.copy_(runner.info, non_blocking=True)
print("two D2H copies + sync: %.3f ms" % tm(copies))
model_cfg = (0.0, 1.25, None)
t0 = time.perf_counter()
for _ in range(50): runner._crop_jobs(1, 0, (0.0, 1.25, np.ones(B)), 0, 2, model_cfg)
torch.cuda.synchronize(); print("host: job table (2 x crop bounds + fields) + H2D enqueue: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
est = np.zeros((B, 5), np.float32); act = np.ones(B, np.int32); pos = np.zeros(B, np.int64)
t0 = time.perf_counter()
for _ in range(50): ops.track_box_by_offset(runner.boxes, est, True, act, pos)
print("host: box update: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))

# ---- the loop itself, instrumented: host time of every segment and the wait for the device ----
import collections
seg = collections.defaultdict(float)
gen = runner._steps(tr)
t_prev = time.perf_counter()
steps = 0
while True:
    try:
        t0 = time.perf_counter(); next(gen); t1 = time.perf_counter()
        seg["host work between sync and yield (enqueue + previous box update)"] += t1 - t0
        runner._done.synchronize(); t2 = time.perf_counter()
        seg["waiting for the device at the yield"] += t2 - t1
        steps += 1
    except StopIteration:
        break
for k, v in seg.items():
    print("%-75s %.3f ms per step" % (k, v / steps * 1e3))

# ---- host cost of each enqueue call of one step (device idle in between) ----
def host(fn, n=30):
    tot = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); tot += time.perf_counter() - t0
    return tot / n * 1e3
print("host enqueue: crop job table + H2D   %.3f ms" % host(lambda: runner._crop_jobs(1, 0, (0.0, 1.25, np.ones(B)), 0, 2, model_cfg)))
print("host enqueue: crop_compact           %.3f ms" % host(lambda: ops.crop_compact(runner.crop_jobs_dev, 2 * B)))
print("host enqueue: regularize             %.3f ms" % host(lambda: ops.regularize(runner.reg_jobs_dev, 2 * B, runner.draws)))
print("host enqueue: graph replay           %.3f ms" % host(lambda: g()))
print("host enqueue: 2 x D2H copy           %.3f ms" % host(copies))
print("host enqueue: event record           %.3f ms" % host(lambda: runner._done.record(torch.cuda.current_stream(dev))))

# ---- one step's host sequence, replicated with timers (device synchronised first) ----
acc = collections.defaultdict(float)
lengths = np.full(B, T)
rng_pos = np.zeros(B, np.int64)
NREP = 25
for i in range(1, 1 + NREP):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    active = (i < lengths).astype(np.int32); t.append(time.perf_counter())
    runner._crop_jobs(i, 0, (0.0, 1.25, np.ones(B) * 2.4), i - 1, 2, model_cfg); t.append(time.perf_counter())
    ops.crop_compact(runner.crop_jobs_dev, 2 * B); ops.regularize(runner.reg_jobs_dev, 2 * B, runner.draws); t.append(time.perf_counter())
    rows = runner._forward(); t.append(time.perf_counter())
    runner.result_host.copy_(rows, non_blocking=True); runner.info_host.copy_(runner.info, non_blocking=True)
    runner._done.record(torch.cuda.current_stream(dev)); t.append(time.perf_counter())
    runner._done.synchronize(); t.append(time.perf_counter())
    est = runner.result_host.numpy(); info = runner.info_host.numpy()
    used = np.where(info[:, 1, 1] > 0, info[:, 1, 1], info[:, 0, 1]); rng_pos = np.where(used > 0, used, rng_pos).astype(np.int64); t.append(time.perf_counter())
    ops.track_box_by_offset(runner.boxes, est, True, active, rng_pos); t.append(time.perf_counter())
    for name, a, b in zip(("active", "crop table", "2 launches", "graph replay", "copies+event", "WAIT", "numpy used", "box update"), t[:-1], t[1:]):
        acc[name] += b - a
for k, v in acc.items():
    print("step segment %-14s %.3f ms" % (k, v / NREP * 1e3))
