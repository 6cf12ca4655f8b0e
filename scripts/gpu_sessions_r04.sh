#!/bin/bash
# Round-4 metered GPU sessions, one function per gpurun call:  gpurun -- 'bash scripts/gpu_sessions_r04.sh <name>'
# Everything is written under gpurun_out/r04<name>/.
set -u
S=${1:?session name}
O=gpurun_out/r04$S
mkdir -p $O
REPO=$(pwd)

ktrace() {   # ktrace <out csv> <cmd...>: rocprofv3 kernel trace + stats of a command, summary copied to $O
    out=$1; shift
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$$ && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$$ -- "$@" \
        > $REPO/$O/$out.stdout 2> $REPO/$O/$out.stderr; f=$(find /tmp/kt_$$ -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/$out.csv)
}

case $S in
a)  # (first session of the round) row-job unit tests, launch-floor probe, the one-tracklet baseline
    timeout 600 python -m pytest tests/test_rowjobs_gpu.py -x -q 2>&1 | tail -15 > $O/rowjobs_test.log
    timeout 600 python scripts/probes/launch_floor_probe.py > $O/launch_floor.log 2>&1
    timeout 300 python scripts/tracklet_b1_profile.py > $O/b1_baseline.log 2>&1
    cat $O/rowjobs_test.log $O/launch_floor.log $O/b1_baseline.log
    ;;
b)  # graph dot dump, parity, one tracklet
    timeout 120 python scripts/probes/graph_dot_probe.py > $O/dot.log 2>&1
    timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/pytest.log
    timeout 300 python scripts/tracklet_b1_profile.py > $O/b1.log 2>&1
    head -60 $O/dot.log; cat $O/pytest.log $O/b1.log
    ;;
c)  # parity, kernel trace of one tracklet
    timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/pytest.log
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    cat $O/pytest.log
    ;;
d)  # the one-frame launch chain on row jobs: parity, then where a tracklet frame's time is
    timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -25 $O/pytest.log
    timeout 300 python scripts/tracklet_b1_profile.py > $O/b1.log 2>&1; grep -v amdgpu.ids $O/b1.log
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    python - <<PY
import csv
rows = list(csv.DictReader(open("$O/b1_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    print("%-100s %6d %8.1fus %5.1f%%" % (r['Name'][:100], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / tot * 100))
PY
    ;;
e)  # row-job kernel after the per-launch specialisation: unit tests, launch-floor probe, one tracklet
    timeout 600 python -m pytest tests/test_rowjobs_gpu.py tests/test_golden_gpu.py tests/test_tracking_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
    timeout 600 python scripts/probes/launch_floor_probe.py > $O/launch_floor.log 2>&1; grep -v amdgpu.ids $O/launch_floor.log
    timeout 300 python scripts/tracklet_b1_profile.py > $O/b1.log 2>&1; grep -v amdgpu.ids $O/b1.log
    ;;
f)  # the default bench line (the driver's command), CPU baseline skipped
    timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
    python - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d["latency_b1"], indent=1))
for k, v in d.get("workloads", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
PY
    ;;
g)  # SA level as row jobs, point jobs: parity, then one tracklet
    timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -25 $O/pytest.log
    timeout 300 python scripts/tracklet_b1_profile.py > $O/b1.log 2>&1; grep -v amdgpu.ids $O/b1.log
    ;;
h)  # kernel trace of one tracklet (per-kernel time and launch count per frame)
    timeout 300 python scripts/tracklet_b1_profile.py > $O/b1.log 2>&1; grep -v amdgpu.ids $O/b1.log
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    python - <<PY
import csv
rows = list(csv.DictReader(open("$O/b1_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
frames = 199 * 5 + 3 + 400
print("device time per frame ~ %.1f us, launches per frame ~ %.1f" % (tot / frames / 1e3, sum(int(r['Calls']) for r in rows) / frames))
for r in rows[:30]:
    print("%-100s %6d %8.1fus %5.1f%% %6.2f/frame" % (r['Name'][:100], int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / tot * 100, int(r['Calls']) / frames))
PY
    ;;
i)  # the review's small parity gaps: shim, frozen BatchNorm, draw-table exhaustion, G15, ped at 2048 + 1024
    timeout 1200 python -m pytest tests/test_round4_gaps_gpu.py tests/test_train_gpu.py tests/test_hot_path_gpu.py -x -q -m gpu -s > $O/pytest.log 2>&1
    grep -E "G15|G10|G14|passed|failed|Error|error" $O/pytest.log | tail -30
    ;;
j)  # split-bf16 GEMM probe beside the exact-fp32 row GEMM on the same shapes
    hipcc --offload-arch=gfx950 -O3 scripts/probes/bf16x3_gemm_probe.hip -o /tmp/bf16x3 2> $O/bf16x3_build.log && timeout 600 /tmp/bf16x3 > $O/bf16x3_probe.log 2>&1
    cat $O/bf16x3_probe.log
    timeout 900 python scripts/rows_gemm_bench.py > $O/rows_gemm.log 2>&1; grep -v amdgpu.ids $O/rows_gemm.log | cut -c1-300
    ;;
z)  # closing evidence of the round: parity tests, the default bench line (all configs), serial / training / one-tracklet kernel
    # traces, PMC passes over the kernels the steps launch NOW (incl. the row jobs and the stress-size pair launch)
    timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
    SECONDS=0; timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall ${SECONDS}s"
    timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
    ktrace serial_kernel_stats python $REPO/bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-workloads
    cp $O/serial_kernel_stats.stdout $O/serial_bench_line.json 2>/dev/null
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 2 --sustain 0     # 22 steps: the first one's one-off launches (first-sight weight packs) weigh 1/22
    cp $O/train_kernel_stats.stdout $O/train_profiled_bench_line.json 2>/dev/null
    bash scripts/train_step_timeline.sh $REPO/$O/train_timeline > $O/train_timeline.log 2>&1; head -1 $O/train_timeline.log     # every launch of ONE step, in order and by shape
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    timeout 300 python scripts/probes/sa_z0_bnbwd_probe.py 2>&1 | grep -v amdgpu.ids > $O/sa_z0_bnbwd_probe.log
    bash scripts/pmc_passes.sh $O/pmc "pair,sa0_s,sa1_s,sa2_s,sa_box,xcorr,lin_,rj" > $O/pmc.log 2>&1; tail -12 $O/pmc.log | cut -c1-300
    bash scripts/pmc_passes.sh $O/pmc_stress "pair" "" "--batch 32 --pair-n 2048,64" > $O/pmc_stress.log 2>&1; tail -4 $O/pmc_stress.log | cut -c1-300
    bash scripts/pmc_passes.sh $O/pmc_train_gemm - "python scripts/rows_gemm_bench.py --no-check --pmc" > $O/pmc_train_gemm.log 2>&1; tail -4 $O/pmc_train_gemm.log | cut -c1-300
    timeout 300 python scripts/train_stream_kernels_bench.py 2>&1 | grep -v amdgpu.ids > $O/train_stream_kernels_bench.log
    bash scripts/pmc_passes.sh $O/pmc_train_stream - "python scripts/train_stream_kernels_bench.py" > $O/pmc_train_stream.log 2>&1; tail -5 $O/pmc_train_stream.log | cut -c1-300
    python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("car", d["value"], d["ms_per_step"], d["roofline"]["frac"], "b1 loop", d["latency_b1"]["tracklet_loop"]["b1"])
for k, v in d.get("workloads", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
PY
    ;;
s)  # stability: the whole GPU suite three times in fresh processes, smoke()
    for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -2; done
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    ;;
t)  # the training step after a change: gradient parity (G10 / G14 / G15, reproducibility), then the step time
    timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_train_config3_gpu.py tests/test_gemm_gpu.py -x -q -m gpu -s > $O/pytest.log 2>&1
    grep -E "G15 \(|G10 on|G14:|passed|failed|Error" $O/pytest.log | tail -12
    timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
    python -c "import json;d=json.load(open('$O/bench_train.json'));print('train', d['ms_per_step'], d['value'], d['sustained'])"
    ;;
u)  # kernel trace of the training step
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 2 --sustain 0
    cp $O/train_kernel_stats.stdout $O/train_profiled_bench_line.json 2>/dev/null
    python - <<PY
import csv
rows = list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 22
ptt = sum(float(r['TotalDurationNs']) for r in rows if 'ptt::' in r['Name'])
print("device ms/step %.2f launches/step %.0f ptt share %.3f" % (tot / steps / 1e6, sum(int(r['Calls']) for r in rows) / steps, ptt / tot))
for r in rows[:45]:
    print("%-110s %6.1f %8.1fus %8.1fus/step" % (r['Name'][:110], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / steps / 1e3))
PY
    ;;
*)  echo "unknown session $S"; exit 2;;
esac
