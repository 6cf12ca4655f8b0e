#!/bin/bash
# Dev tool (GPU box): per-dispatch kernel trace of a short training run. Writes (1) every launch of the LAST step in launch order
# (name, grid, workgroup, duration) — the neighbours of an ATen launch name its call site, which torch.profiler's stacks do not
# for the backward thread — and (2) the launches grouped by (kernel, grid) with count and average duration.
O=${1:-$GRAFT_REPO_ROOT/gpurun_out/timeline}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 3 --warmup 2 --no-cpu-baseline --sustain 0 --no-extras > $O/bench_line.json 2> $O/bench_err.log
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" "$O" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'at::native::\(anonymous namespace\)::', 'at::', n)
    n = re.sub(r'void ', '', n)
    return n[:110]
# steps are delimited by the optimizer's update launch
marks = [i for i, r in enumerate(rows) if 'adam_update_kernel' in r['Kernel_Name']]
step_end = []
for i in marks:
    if not step_end or i - step_end[-1] > 50: step_end.append(i)
    else: step_end[-1] = i
lo, hi = (step_end[-2] + 1, step_end[-1] + 1) if len(step_end) >= 2 else (0, len(rows))
with open(sys.argv[2] + '/last_step_launches.txt', 'w') as f:
    t0 = int(rows[lo]['Start_Timestamp'])
    for k, r in enumerate(rows[lo:hi]):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        f.write("%4d %9.1f %8.1f us  g=%sx%sx%s wg=%s  %s\n" % (k, (int(r['Start_Timestamp']) - t0) / 1e3, d, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'],
                                                         r['Workgroup_Size_X'], short(r['Kernel_Name'])))
print("last step: %d launches, %.2f ms of kernels" % (hi - lo, sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[lo:hi]) / 1e6))
agg = collections.defaultdict(list)
for r in rows[lo:hi]:
    agg[(short(r['Kernel_Name'])[:90], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(sys.argv[2] + '/last_step_by_shape.txt', 'w') as f:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%-92s grid %8s x %4s wg %4s  n=%3d avg %8.1f us total %8.1f\n" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), sum(v)))
PY
head -50 $O/last_step_by_shape.txt | cut -c1-200
