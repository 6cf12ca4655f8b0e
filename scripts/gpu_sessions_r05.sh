#!/bin/bash
# Round-5 metered GPU sessions, one function per gpurun call:  gpurun -- 'bash scripts/gpu_sessions_r05.sh <name>'
# Everything is written under gpurun_out/r05<name>/.
set -u
S=${1:?session name}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05$S
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
a)  # the round's first changes (advisor fixes, workspace FPS, one-rank RCCL checks, compact bench line): parity, then the suite
    timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_bench_gpu.py tests/test_tracking_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
    timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
    ;;
c)  # the in-process car -> ped -> stress crash in hipGraphLaunch: reproduction, queue bookkeeping of the
    # crashing (A) and a passing (D) order, the pure-PyTorch reproduction under 1 / 2 / 4 hardware queues, the mitigation
    probe A.log PROBE_X=1 timeout 300 python scripts/probes/graph_sequence_probe.py bench car,ped,stress
    probe D.log PROBE_X=1 timeout 300 python scripts/probes/graph_sequence_probe.py bench ped,car,stress
    PAT='Selected queue|hipGraphInstantiate|hipGraphLaunch \(|hipGraphExecDestroy|hipStreamCreate|hardware queues|\[probe\]|Fatal'
    for t in A:car,ped,stress D:ped,car,stress; do
        AMD_LOG_LEVEL=3 timeout 600 python scripts/probes/graph_sequence_probe.py bench ${t##*:} > /tmp/q.log 2>&1
        grep -E "$PAT" /tmp/q.log | cut -c1-220 | tail -2500 > $O/queues_${t%%:*}.log
    done
    for q in 1 2 4; do for b in 2 3; do probe repro_q${q}_b$b.log GPU_MAX_HW_QUEUES=$q timeout 120 python scripts/probes/graph_queue_repro.py $b 0 drop null; done; done
    for q in 1 2; do for d in graphed pipelined; do for o in 0 1; do
        probe drop_q${q}_${d}_overlap$o.log GPU_MAX_HW_QUEUES=$q PROBE_DRIVER=$d PROBE_OVERLAP=$o timeout 200 python scripts/probes/graph_sequence_probe.py drop car
    done; done; done
    probe mitigation.log GPU_MAX_HW_QUEUES=8 PROBE_ROUNDS=7 timeout 1200 python scripts/probes/graph_sequence_probe.py bench car,ped,stress
    probe mitigation_train.log GPU_MAX_HW_QUEUES=8 PROBE_ROUNDS=3 timeout 1200 python scripts/probes/graph_sequence_probe.py bench car,ped,stress,train
    ;;
e)  # what 8 hardware queues cost: one tracklet frame and the headline with GPU_MAX_HW_QUEUES = 4 / 8; the ns = 1 gradient diagnostic
    timeout 300 python scripts/probes/ns1_grad_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|errs =" | tee $O/ns1_diag.log | cut -c1-330
    for q in 4 8 4 8; do echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 300 python scripts/tracklet_b1_profile.py 2>&1 | grep -v amdgpu.ids; done | tee $O/b1_queues.log
    for q in 4 8; do GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-workloads 2> $O/bench_q$q.err > $O/bench_q$q.json; done
    ;;
h)  # ptt_rows_gemm_bnbwd_fused_f32: the dz it writes out (store hazard), the three-step diagnostic, the suite, the step with / without it
    timeout 600 python scripts/probes/fused_bnbwd_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/fused_probe.log | cut -c1-300
    timeout 600 python scripts/probes/three_step_diag.py 2>&1 | grep -v amdgpu.ids | tee $O/three_step.log | cut -c1-200
    timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "^FAILED|passed|failed" $O/pytest_gpu.log | cut -c1-200
    for m in 1 0 1; do
        PTT_FUSED_BN_BWD=$m timeout 300 python bench.py --workload train --steps 20 --warmup 5 --sustain 2 --no-cpu-baseline 2> $O/train_f$m.err | tee -a $O/train_modes.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $m', d['ms_per_step'], d['sustained'])"
    done
    ;;
k)  # the gradient sink (one finishing launch per step): its tests, the training tests, the step with / without it
    timeout 900 python -m pytest tests/test_grad_sink_gpu.py -x -q -m gpu -s > $O/pytest_sink.log 2>&1; grep -E "flat gradient|passed|failed|Error|error" $O/pytest_sink.log | tail -12 | cut -c1-300
    timeout 1500 python -m pytest tests/test_train_config3_gpu.py tests/test_train_gpu.py tests/test_step_ops_gpu.py tests/test_syncbn_gpu.py tests/test_bench_gpu.py -q -m gpu -s > $O/pytest_train.log 2>&1
    grep -E "^FAILED|passed|failed|worst gradient" $O/pytest_train.log | tail -12 | cut -c1-300
    for m in flat ddp flat; do
        PTT_TRAIN_REDUCER=$m timeout 300 python bench.py --workload train --steps 20 --warmup 5 --sustain 2 --no-cpu-baseline 2> $O/train_$m.err | tee -a $O/train_modes.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reducer $m', d['ms_per_step'], d['sustained'])"
    done
    ;;
m)  # the fused attention core (_AttnCore): its tests, the training tests, the step with / without it
    timeout 900 python -m pytest tests/test_attn_core_gpu.py -x -q -m gpu -s > $O/pytest_core.log 2>&1; grep -E "fused attention|passed|failed|Error|error" $O/pytest_core.log | tail -12 | cut -c1-300
    timeout 1500 python -m pytest tests/test_train_config3_gpu.py tests/test_train_gpu.py tests/test_step_ops_gpu.py tests/test_grad_sink_gpu.py -q -m gpu -s > $O/pytest_train.log 2>&1
    grep -E "^FAILED|passed|failed|G10 on|G15 \(|G14:" $O/pytest_train.log | tail -12 | cut -c1-300
    for m in 1 0 1; do
        PTT_ATTN_CORE=$m timeout 300 python bench.py --workload train --steps 20 --warmup 5 --sustain 2 --no-cpu-baseline 2> $O/train_core$m.err | tee -a $O/train_modes.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attention core $m', d['ms_per_step'], d['sustained'])"
    done
    ;;
g)  # weight-gradient tile shapes (needs a build with PTT_GEMM_FLAGS=-DPTT_GEMM_DEV)
    WG_FIRSTS=0,1,2,3 timeout 600 python scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_bench.log
    ;;
s)  # the pair kernel in Morton order at the stress size: parity, kernel time and HBM traffic with / without the order, the workload
    timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_hot_path_gpu.py tests/test_golden_gpu.py -q -m gpu > $O/pytest.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-200
    for o in none spatial none spatial; do PAIR_ORDER=$o timeout 300 python scripts/kernel_bench.py --only pair --batch 32 --pair-n 2048,64 --iters 10 2>&1 | grep pair_N2048 | sed "s/^/$o /"; done | tee $O/pair_order.log
    PAIR_ORDER=none bash scripts/pmc_passes.sh $O/pmc_stress_sampling_order "pair" "" "--batch 32 --pair-n 2048,64" > $O/pmc_none.log 2>&1; tail -3 $O/pmc_none.log | cut -c1-400
    bash scripts/pmc_passes.sh $O/pmc_stress "pair" "" "--batch 32 --pair-n 2048,64" > $O/pmc_spatial.log 2>&1; tail -3 $O/pmc_spatial.log | cut -c1-400
    for i in 1 2; do timeout 600 python bench.py --workload stress --steps 8 --warmup 3 --no-cpu-baseline --sustain 0 2> $O/stress.err | tee $O/stress.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stress', d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernel_ms_per_step'])"; done
    ;;
q)  # the uniform-grid ball query: index parity (sweep, oracle), then the stress and car workloads
    timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_point_ops_gpu.py tests/test_golden_gpu.py tests/test_hot_path_gpu.py tests/test_tracking_gpu.py -q -m gpu > $O/pytest.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-200
    for w in stress stress car; do timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-workloads --no-full-model --no-latency 2> $O/$w.err | tee $O/$w.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"; done
    ;;
t)  # the training step after a change: gradient parity (G10 / G14 / G15, reproducibility, weight-gradient kernels), then the step time
    timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_train_config3_gpu.py tests/test_gemm_gpu.py tests/test_step_ops_gpu.py tests/test_round5_gpu.py -q -m gpu > $O/pytest.log 2>&1
    grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-200
    for i in 1 2; do timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench_train.err | tee $O/bench_train.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['value'], d['sustained'])"; done
    ;;
u)  # kernel trace of the training step, per-dispatch timeline of its last step
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 2 --sustain 0 --no-cpu-baseline
    bash scripts/train_step_timeline.sh $REPO/$O/timeline > $O/timeline.log 2>&1; tail -3 $O/timeline.log
    python - <<PY
import csv
rows = list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = sum(int(r['Calls']) for r in rows)
ptt = sum(float(r['TotalDurationNs']) for r in rows if 'ptt::' in r['Name'])
print("train: %.2f ms of kernels per step, %.1f launches per step, %.1f %% ptt::" % (tot / 22 / 1e6, n / 22, 100 * ptt / tot))
for r in rows[:30]:
    print("%-100s %6.1f/step %8.1fus %6.3f ms/step" % (r['Name'][:100], int(r['Calls']) / 22, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 22 / 1e6))
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
    ktrace train_kernel_stats python $REPO/bench.py --workload train --steps 20 --warmup 2 --sustain 0 --no-cpu-baseline
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
