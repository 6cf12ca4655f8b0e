#!/bin/bash
# Dev tool (GPU box): per-dispatch kernel trace of the eager one-stream headline pass (bench.py --serial); the launches of the LAST
# step grouped by (kernel, grid) with workgroup counts — which launches leave CUs idle.
O=${1:-$GRAFT_REPO_ROOT/gpurun_out/serial_by_shape}; W=${2:-car}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kts && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kts -- python $GRAFT_REPO_ROOT/bench.py --workload $W --serial --steps 4 --warmup 2 --sustain 0 --no-cpu-baseline --no-workloads --no-full-model --no-latency > $O/line.json 2> $O/err.log
f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
python - "$f" "$O" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'fps_kernel' in r['Kernel_Name'] and int(r['Grid_Size_X']) >= 48 * 256 and '256, 8' in r['Kernel_Name'] or ('fps_kernel' in r['Kernel_Name'] and int(r['Grid_Size_X']) > 16000)]
lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(rows))
agg = collections.defaultdict(list)
for r in rows[lo:hi]:
    n = re.sub(r'void |ptt::', '', r['Kernel_Name'])[:80]
    wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']))
    agg[(n, wgs)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(sys.argv[2] + '/by_shape.txt', 'w') as f:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f.write("%-82s wgs %6d n=%3d avg %8.1f us total %8.1f\n" % (k[0], k[1], len(v), sum(v) / len(v), sum(v)))
    f.write("step: %d launches, %.1f us of kernels\n" % (hi - lo, sum(sum(v) for v in agg.values())))
print(open(sys.argv[2] + '/by_shape.txt').read())
PY
