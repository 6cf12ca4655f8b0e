#!/bin/bash
# Round 5, session b: weight gradients on a second stream (PTT_WGRAD_STREAM 0 / 1 / 2), wgrad2 tiles of a row chunk on one XCD;
# the in-process bench sequence probe.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_train_config3_gpu.py tests/test_step_ops_gpu.py tests/test_gemm_gpu.py tests/test_golden_gpu.py tests/test_syncbn_gpu.py -x -q -m gpu > $O/pytest_train.log 2>&1; echo "rc=$?" >> $O/pytest_train.log
tail -3 $O/pytest_train.log
for m in 0 1 2 0 2; do
  PTT_WGRAD_STREAM=$m timeout 300 python bench.py --workload train --steps 20 --warmup 5 --sustain 2 --no-cpu-baseline 2> $O/train_m$m.err | tee -a $O/train_modes.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $m', d['ms_per_step'], d['sustained'])"
done
PROBE_ROUNDS=2 timeout 600 python scripts/probes/graph_sequence_probe.py bench car,ped,stress,train > $O/probe_bench.log 2>&1; echo "rc=$?" >> $O/probe_bench.log
tail -4 $O/probe_bench.log
