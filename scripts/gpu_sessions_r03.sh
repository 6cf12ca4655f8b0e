#!/bin/bash
# Round-3 metered GPU sessions, one function per gpurun call:  gpurun -- 'bash scripts/gpu_sessions_r03.sh <name>'
# (round 2's closing session is scripts/gpu_session_r02z.sh; the per-session scripts of rounds 1-2 are summarised in
# profiles/README.md). Everything is written under gpurun_out/<name>/.
set -u
S=${1:?session name}
O=gpurun_out/r03$S
mkdir -p $O
REPO=$(pwd)

ktrace() {   # ktrace <out csv> <cmd...>: rocprofv3 kernel trace + stats of a command, summary copied to $O
    out=$1; shift
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt_$$ && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$$ -- "$@" \
        > $REPO/$O/$out.stdout 2> $REPO/$O/$out.stderr; f=$(find /tmp/kt_$$ -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $REPO/$O/$out.csv)
}

case $S in
a)  # configs[3] under test on the GPU (world 1 at B = 48, two-rank DDP on the row kernels, sync_bn) + GEMM baselines
    timeout 900 python -m pytest tests/test_train_config3_gpu.py -m gpu -x -q -s > $O/pytest_config3.log 2>&1; echo "rc=$?" >> $O/pytest_config3.log
    tail -25 $O/pytest_config3.log
    timeout 600 python scripts/rows_gemm_bench.py > $O/rows_gemm.log 2>&1; cat $O/rows_gemm.log
    timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; tail -c 600 $O/bench_train.json
    ;;
b)  timeout 1200 python -m pytest tests/test_train_config3_gpu.py -m gpu -q -s > $O/pytest_config3.log 2>&1; echo "rc=$?" >> $O/pytest_config3.log
    tail -40 $O/pytest_config3.log
    ;;
c)  timeout 900 python scripts/rows_gemm_bench.py > $O/rows_gemm.log 2>&1; cat $O/rows_gemm.log | cut -c1-420
    ;;
d)  timeout 900 python scripts/rows_gemm_exp.py > $O/rows_gemm_exp.log 2>&1; cat $O/rows_gemm_exp.log | cut -c1-420
    ;;
e)  hipcc --offload-arch=gfx950 -O2 scripts/probes/buffer_range_probe.hip -o /tmp/brp && /tmp/brp > $O/buffer_range_probe.log 2>&1; cat $O/buffer_range_probe.log
    timeout 900 python scripts/rows_gemm_bench.py > $O/rows_gemm.log 2>&1; grep -v amdgpu.ids $O/rows_gemm.log | cut -c1-420
    timeout 900 python scripts/rows_gemm_exp.py > $O/rows_gemm_exp.log 2>&1; grep -v amdgpu.ids $O/rows_gemm_exp.log | cut -c1-420
    ;;
f)  # the training path on the persistent row GEMM: parity tests, then the step
    timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_syncbn_gpu.py -m gpu -q -x > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
    tail -5 $O/pytest_train.log
    timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; python -c "import json;d=json.load(open('$O/bench_train.json'));print(d['ms_per_step'],d['value'],d['sustained'])"
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 5 --warmup 2 --sustain 0
    ;;
g)  timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
    tail -30 $O/pytest_gpu.log
    ;;
h)  # the default driver command
    SECONDS=0; PTT_BENCH_VERBOSE=1 timeout 1500 python -X faulthandler bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; grep -i "elapsed\|error\|Traceback" $O/bench_default.err | head
    ;;
i)  # isolate the stress-after-car crash
    for w in stress ped,stress; do
        PTT_BENCH_VERBOSE=1 timeout 600 python -X faulthandler bench.py --no-cpu-baseline --no-full-model --no-latency --steps 5 --warmup 2 --sustain 0 --workloads $w > $O/b_$w.json 2> $O/b_$w.err
        echo "== $w rc=$?"; grep -v amdgpu.ids $O/b_$w.err | head -12 | cut -c1-160
    done
    PTT_BENCH_VERBOSE=1 timeout 600 python -X faulthandler bench.py --workload stress --no-cpu-baseline --steps 5 --warmup 2 --sustain 0 > $O/b_stress_alone.json 2> $O/b_stress_alone.err
    echo "== stress alone rc=$?"; grep -v amdgpu.ids $O/b_stress_alone.err | head -12 | cut -c1-160
    ;;
j)  # which stage poisons the later stress graph: full model + latency, or the CPU baseline
    PTT_BENCH_VERBOSE=1 timeout 900 python -X faulthandler bench.py --no-cpu-baseline --workloads stress > $O/b_nocpu.json 2> $O/b_nocpu.err
    echo "== no cpu baseline rc=$?"; grep -v amdgpu.ids $O/b_nocpu.err | head -14 | cut -c1-160
    PTT_BENCH_VERBOSE=1 timeout 900 python -X faulthandler bench.py --no-full-model --no-latency --workloads stress > $O/b_cpu.json 2> $O/b_cpu.err
    echo "== cpu baseline only rc=$?"; grep -v amdgpu.ids $O/b_cpu.err | head -14 | cut -c1-160
    nproc; python -c "import os;print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
    ;;
k)  true
    true
    SECONDS=0; PTT_BENCH_VERBOSE=1 timeout 1500 python -X faulthandler bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "== default rc=$? wall ${SECONDS}s"; grep -v amdgpu.ids $O/bench_default.err | head -30 | cut -c1-160
    ;;
l)  timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_golden_gpu.py -m gpu -q -s -k "G10 or G14 or hoisted or transformer_block_training or shared_mlp_pool or conv1d_stack" > $O/pytest_tol.log 2>&1; echo "rc=$?" >> $O/pytest_tol.log
    grep -i "worst\|measured\|passed\|failed\|observed" $O/pytest_tol.log | cut -c1-300
    ;;
m)  for m in rows stock; do for f in 1; do echo "force picks $f"; G14_FORCE_PICKS=$f G14_PATH=$m timeout 600 python scripts/g14_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|return float" | cut -c1-200; done; done > $O/g14_diag.log; cat $O/g14_diag.log
    ;;
n)  timeout 600 python scripts/rows_gemm_bench.py --no-bench 2>&1 | grep check | cut -c1-250
    G14_ROWS=2 timeout 600 python scripts/g14_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|return float\|group" | cut -c1-200
    true python scripts/fwd_precision_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/fwd_precision.log; cat $O/fwd_precision.log
    ;;
o)  # one tracklet frame at B = 1: the launch chain
    timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_dense_gpu.py tests/test_hot_path_gpu.py tests/test_tracking_gpu.py -m gpu -q -x 2>&1 | tail -40 | cut -c1-200
    timeout 300 python scripts/tracklet_b1_profile.py 2>&1 | grep -v amdgpu.ids > $O/b1.log; cat $O/b1.log
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03o/b1_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
nfr=200+4+200+200   # frames replayed by the script (runner.run twice + two replay loops)
print("device time %.1f ms total; per frame ~%.3f ms over %d frames; launches per frame %.1f" % (tot/1e6, tot/1e6/nfr, nfr, sum(int(r['Calls']) for r in rows)/nfr))
for r in rows[:32]:
    print("%-80s %6s %8.1f us  %6.1f us/frame" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/nfr))
PY
    ;;
p)  timeout 600 python scripts/linear_infer_bench.py 2>&1 | grep -v amdgpu.ids > $O/linear_infer.log; cat $O/linear_infer.log
    ;;
q)  timeout 600 python scripts/train_aten_ops.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/train_aten_ops.log; cat $O/train_aten_ops.log | cut -c1-230
    ;;
t)  timeout 600 python scripts/train_small_ops.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/train_small_ops.log; cat $O/train_small_ops.log | cut -c1-230
    ;;
w)  # launch consolidation of the training step: batched weight packing, BatchNorm bookkeeping inside the statistics launch
    timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_train_gpu.py tests/test_train_config3_gpu.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-250
    timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --sustain 0
    python - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("launches/step %.0f  device ms/step %.2f" % (calls/7, tot/7e6))
for r in rows[:60]:
    if int(r['Calls'])/7 >= 20: print("%7.1f/step %7.3f ms/step  %s" % (int(r['Calls'])/7, float(r['TotalDurationNs'])/7e6, r['Name'][:110]))
PY
    ;;
z)  # closing evidence of the round: parity tests, the default bench line (all configs), serial + training kernel traces, PMC passes
    timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
    SECONDS=0; timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$? wall ${SECONDS}s"
    timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
    ktrace serial_kernel_stats python $REPO/bench.py --serial --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-workloads
    cp $O/serial_kernel_stats.stdout $O/serial_bench_line.json 2>/dev/null
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 5 --warmup 2 --sustain 0
    cp $O/train_kernel_stats.stdout $O/train_profiled_bench_line.json 2>/dev/null
    ktrace b1_kernel_stats python $REPO/scripts/tracklet_b1_profile.py
    bash scripts/pmc_passes.sh $O/pmc "pair,sa0_s,sa1_s,sa2_s,sa_box" > $O/pmc.log 2>&1; tail -8 $O/pmc.log | cut -c1-300
    bash scripts/pmc_passes.sh $O/pmc_train_gemm - "python scripts/rows_gemm_bench.py --no-check --pmc" > $O/pmc_train_gemm.log 2>&1; tail -8 $O/pmc_train_gemm.log | cut -c1-300
    timeout 600 python scripts/fwd_precision_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/forward_precision_vs_float64.log; tail -12 $O/forward_precision_vs_float64.log
    G14_ROWS=8 timeout 600 python scripts/g14_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|return float" > $O/gradient_error_vs_float64.log; tail -4 $O/gradient_error_vs_float64.log
    ;;
r)  timeout 900 python -m pytest tests/test_tracking_gpu.py -m gpu -q 2>&1 | tail -15 | cut -c1-200
    ;;
s)  # stability: the whole GPU suite three times in fresh processes, smoke() and build()
    for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -2; done
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
    ;;
*)  echo "unknown session $S"; exit 2;;
esac
