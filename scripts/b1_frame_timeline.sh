#!/bin/bash
# Dev tool (GPU box): per-dispatch kernel trace of the one-tracklet loop; prints ONE frame's launches in start order with their
# queue, start offset, duration and the gap to the previous end on the same queue -> where the frame's critical path waits.
O=${1:-$GRAFT_REPO_ROOT/gpurun_out/b1_timeline}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ktb && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktb -- python $GRAFT_REPO_ROOT/scripts/tracklet_b1_profile.py > $O/run.log 2>&1
f=$(find /tmp/ktb -name "*kernel_trace.csv" | head -1)
python - "$f" "$O" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'crop_regularize_kernel' in r['Kernel_Name']]
lo, hi = marks[-3], marks[-2]                      # one whole frame late in the run
t0 = int(rows[lo]['Start_Timestamp'])
last_end = {}
with open(sys.argv[2] + '/one_frame.txt', 'w') as f:
    for r in rows[lo:hi]:
        q = r.get('Queue_Id', '?')
        st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        gap = st - last_end.get(q, st)
        last_end[q] = en
        name = re.sub(r'void |ptt::', '', r['Kernel_Name'])[:70]
        f.write("q%-3s start %8.1f us  dur %7.1f us  gap %6.1f us  g=%s wg=%s  %s\n" % (q, st / 1e3, (en - st) / 1e3, gap / 1e3, r['Grid_Size_X'], r['Workgroup_Size_X'], name))
    f.write("frame: %d launches, span %.1f us\n" % (hi - lo, (int(rows[hi - 1]['End_Timestamp']) - t0) / 1e3))
print(open(sys.argv[2] + '/one_frame.txt').read())
PY
