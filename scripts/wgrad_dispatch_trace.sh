#!/bin/bash
# Dev tool (GPU box): per-dispatch kernel trace of a short training run, the weight-gradient / linear launches grouped by grid —
# which shapes land on which kernel and what each costs (found SA0's 64-channel layers on zero-padded 128 x 128 blocks).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 3 --warmup 2 --no-cpu-baseline --sustain 0 > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'linear_wgrad_kernel' in n or 'wgrad_finish' in n or 'wgrad2' in n or 'linear_kernel' in n:
        key=(n[:60], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
        agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print("%-62s grid %7s x %4s wg %4s  n=%3d avg %8.1f us total %8.1f" % (k[0],k[1],k[2],k[3],len(v),sum(v)/len(v),sum(v)))
PY
