#!/bin/bash
set -u
O=gpurun_out/r02j
mkdir -p $O
REPO=$(pwd)
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -s 2>&1 | grep -E "G10 on|passed|failed|Error|error" | head
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktt -- \
    python $REPO/bench.py --workload train --steps 5 --warmup 2 --sustain 0 > $REPO/$O/train_prof.json 2> $REPO/$O/train_prof.err; \
    f=$(find /tmp/ktt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/train_kernel_stats.csv)
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms per step:", tot/1e6/7)
for r in rows[:22]:
    print("%6.2f%% %5s calls %9.1f us avg  %s" % (float(r['Percentage']), r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:100]))
d=json.loads(open("$O/train_prof.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])
PY
