#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O; REPO=$(pwd)
timeout 300 python scripts/tracklet_b1_profile.py 2>&1 | grep -v amdgpu
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kb1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb1 -- python $REPO/scripts/tracklet_b1_profile.py > /dev/null 2>&1; f=$(find /tmp/kb1 -name "*kernel_stats.csv" | head -1); cp $f $REPO/$O/b1_kernel_stats.csv)
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/b1_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("device kernel time total %.1f ms over ~%d frames" % (tot/1e6, 203+400))
for r in rows[:16]: print("%6.2f%% %6s calls %8.1f us avg %s" % (float(r['Percentage']), r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:80]))
PY
