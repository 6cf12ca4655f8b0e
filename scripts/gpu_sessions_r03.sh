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
e)  hipcc --offload-arch=gfx950 -O2 scripts/buffer_range_probe.hip -o /tmp/brp && /tmp/brp > $O/buffer_range_probe.log 2>&1; cat $O/buffer_range_probe.log
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
*)  echo "unknown session $S"; exit 2;;
esac
