"""Dev tool: from a rocprofv3 --kernel-trace CSV of `bench.py` (graphed, pipelined step) — how the step's wall time
splits into matrix kernels, index kernels and idle, with the kernels as they run IN the step (concurrent streams)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
# the timed region = the last third of the trace (warm-up and graph capture come first)
t0 = ks[len(ks) * 2 // 3][0]; t1 = ks[-1][1]
sel = [k for k in ks if k[0] >= t0]
wall = (t1 - t0) / 1e3
def cls(n):
    if "pt_attn_pair" in n: return "pair"
    if "sa_stream" in n or "sa_lds" in n or "sa_fused" in n: return "sa"
    if "linear_kernel" in n: return "linear"
    if "fps_kernel" in n: return "fps"
    if "ball_query" in n or "knn" in n: return "index"
    return "other"
dur = collections.Counter(); cnt = collections.Counter()
for s, e, n in sel:
    dur[cls(n)] += (e - s) / 1e3; cnt[cls(n)] += 1
# union coverage of the matrix kernels, and of everything
def union(items):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(items):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot / 1e3
mat = [(s, e) for s, e, n in sel if cls(n) in ("pair", "sa", "linear")]
npair = cnt["pair"]
steps = npair / 2.0
print("window %.1f us, %d kernels, ~%.1f steps -> %.1f us per step" % (wall, len(sel), steps, wall / steps))
for k in ("pair", "sa", "linear", "fps", "index", "other"):
    print("  %-7s %5d launches  %9.1f us total  %8.1f us per step" % (k, cnt[k], dur[k], dur[k] / steps))
print("  matrix kernels: sum %.1f us per step, union (time at least one runs) %.1f us per step" % (
    (dur["pair"] + dur["sa"] + dur["linear"]) / steps, union(mat) / steps))
print("  any kernel running: %.1f us per step" % (union([(s, e) for s, e, n in sel]) / steps))
