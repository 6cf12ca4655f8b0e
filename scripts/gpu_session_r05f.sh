#!/bin/bash
# Round 5, session f: the whole -m gpu suite, the default bench line, the kernel trace of the training step (fused BatchNorm backward)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05f; mkdir -p $O; REPO=$(pwd)
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? line $(wc -c < $O/bench_default.json) bytes"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py --workload train --steps 20 --warmup 2 --sustain 0 --no-cpu-baseline > $REPO/$O/train_profiled_line.json 2> $REPO/$O/train_profiled.err; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/train_kernel_stats.csv)
bash scripts/train_step_timeline.sh $REPO/$O/timeline > $O/timeline.log 2>&1; tail -3 $O/timeline.log
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
ptt = sum(float(r['TotalDurationNs']) for r in rows if 'ptt::' in r['Name'])
print("train: %.2f ms of kernels per step, %.1f launches per step, %.1f %% ptt::" % (tot / 22 / 1e6, n / 22, 100 * ptt / tot))
for r in rows[:28]:
    print("%-100s %6.1f/step %8.1fus %6.3f ms/step" % (r['Name'][:100], int(r['Calls']) / 22, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 22 / 1e6))
d = json.load(open("$O/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["latency_b1"]["tracklet_loop"]["b1"], {k: (v.get("ms_per_step"), v.get("error")) for k, v in d["workloads"].items()})
PY
