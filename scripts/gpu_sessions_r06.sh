#!/bin/bash
# Round-6 metered GPU sessions, one function per gpurun call:  gpurun -- 'bash scripts/gpu_sessions_r06.sh <name>'
# Everything is written under gpurun_out/r05<name>/.
set -u
S=${1:?session name}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06$S
mkdir -p $O
REPO=$(pwd)

ktrace() {   # ktrace <out csv> <cmd...>: rocprofv3 kernel trace + stats of a command, summary copied to $O
    out=$1; shift
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$$ && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$$ -- "$@" \
        > $REPO/$O/$out.stdout 2> $REPO/$O/$out.stderr; f=$(find /tmp/kt_$$ -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/$out.csv)
}
probe() {    # probe <log> ENV=.. -- mode order: scripts/probes/graph_sequence_probe.py in its own process, exit code appended
    log=$1; shift
    env "$@" > $O/$log 2>&1; echo "rc=$?" >> $O/$log; echo "== $log: $(grep -E 'PASSED|Fatal' $O/$log | tail -1) $(tail -1 $O/$log)"
}

case $S in
a)  # first evidence of the round: the default bench line, the training line on the captured step, its kernel trace
    cat ptt_amd/lib/BUILD_ID
    timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? line $(wc -c < $O/bench_default.json) bytes"
    timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err; cat $O/bench_train.json | cut -c1-1500
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-extras
    cp $O/train_kernel_stats.stdout $O/train_profiled_bench_line.json
    python - <<PY
import csv, json
d = json.load(open("$O/bench_default.json"))
print("car", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "b1", d["latency_b1"]["tracklet_loop"]["b1"]["ms_per_step"], "whole_step", d.get("whole_step"))
print({k: (v.get("ms_per_step"), v.get("host_issue_ms_per_step"), v.get("small_batch")) for k, v in d["workloads"].items()})
print("affinity", d.get("cpu_affinity"), "cpu", d["cpu_baseline"])
rows = list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
ptt = sum(float(r['TotalDurationNs']) for r in rows if 'ptt::' in r['Name'])
print("train: %.2f ms of kernels per step, %.1f launches per step, %.1f %% ptt::" % (tot / 25 / 1e6, n / 25, 100 * ptt / tot))
for r in rows[:40]:
    print("%-100s %6.1f/step %8.1fus %6.3f ms/step" % (r['Name'][:100], int(r['Calls']) / 25, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 25 / 1e6))
PY
    ;;
t)  # the training step after a change: gradient parity (G10 / G14 / G15, reproducibility, weight-gradient kernels), then the step time
    timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_train_config3_gpu.py tests/test_gemm_gpu.py tests/test_step_ops_gpu.py tests/test_round5_gpu.py -q -m gpu > $O/pytest.log 2>&1
    grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-200
    for i in 1 2; do timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_train.err | tee $O/bench_train.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['value'], d['sustained'])"; done
    ;;
u)  # kernel trace of the training step, per-dispatch timeline of its last step
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-extras
    bash scripts/train_step_timeline.sh $REPO/$O/timeline > $O/timeline.log 2>&1; tail -3 $O/timeline.log
    python - <<PY
import csv
rows = list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
ptt = sum(float(r['TotalDurationNs']) for r in rows if 'ptt::' in r['Name'])
print("train: %.2f ms of kernels per step, %.1f launches per step, %.1f %% ptt::" % (tot / 25 / 1e6, n / 25, 100 * ptt / tot))
for r in rows[:30]:
    print("%-100s %6.1f/step %8.1fus %6.3f ms/step" % (r['Name'][:100], int(r['Calls']) / 25, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 25 / 1e6))
PY
    ;;
z)  # closing evidence of the round on a committed revision: parity, the default bench line, serial / training / one-tracklet kernel
    # traces, PMC passes (ptt_amd/lib/BUILD_ID names the revision)
    cat ptt_amd/lib/BUILD_ID
    timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
    timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? line $(wc -c < $O/bench_default.json) bytes"
    timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
    ktrace serial_kernel_stats python $REPO/bench.py --serial --steps 10 --warmup 3 --sustain 0 --no-cpu-baseline --no-workloads --no-full-model --no-latency
    cp $O/serial_kernel_stats.stdout $O/serial_bench_line.json
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-extras
    cp $O/train_kernel_stats.stdout $O/train_profiled_bench_line.json
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    bash scripts/train_step_timeline.sh $REPO/$O/train_timeline > $O/train_timeline.log 2>&1
    ;;
p)  # PMC passes (separate counter-only runs, scripts/pmc_passes.sh): car pair kernel, stress pair kernel, the training GEMMs
    cat ptt_amd/lib/BUILD_ID
    bash scripts/pmc_passes.sh $O/pmc "pair,sa0_s,sa1_s,sa2_s,sa_box,xcorr,lin_,rj" > $O/pmc.log 2>&1; tail -12 $O/pmc.log | cut -c1-300
    bash scripts/pmc_passes.sh $O/pmc_stress "pair" "" "--batch 32 --pair-n 2048,64" > $O/pmc_stress.log 2>&1; tail -4 $O/pmc_stress.log | cut -c1-300
    PAIR_ORDER=none bash scripts/pmc_passes.sh $O/pmc_stress_sampling_order "pair" "" "--batch 32 --pair-n 2048,64" > $O/pmc_stress_sampling_order.log 2>&1
    bash scripts/pmc_passes.sh $O/pmc_train_gemm - "python scripts/rows_gemm_bench.py --no-check --pmc" > $O/pmc_train_gemm.log 2>&1; tail -4 $O/pmc_train_gemm.log | cut -c1-300
    bash scripts/pmc_train_step.sh $O/pmc_train_step > $O/pmc_train_step.log 2>&1; tail -2 $O/pmc_train_step.log
    ;;
y)  # stability of the closing build: the GPU suite three times in fresh processes, smoke(), the default bench line twice
    for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done | tee $O/suite_x3.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
    for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['latency_b1']['tracklet_loop']['b1']['ms_per_step'], {k: v.get('ms_per_step') for k, v in d['workloads'].items()})"; done | tee $O/bench_x2.log
    ;;
*)  echo "unknown session $S"; exit 2 ;;
esac
