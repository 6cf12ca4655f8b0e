#!/bin/bash
# Round 5, session a: the round's first changes on the GPU — new tests, the whole -m gpu suite, the graph-sequence probe.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/env.log 2>&1
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_bench_gpu.py tests/test_tracking_gpu.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/pytest_new.log
tail -5 $O/pytest_new.log
for mode in drop alive empty eager; do
  timeout 300 python scripts/probes/graph_sequence_probe.py $mode > $O/probe_$mode.log 2>&1; echo "rc=$?" >> $O/probe_$mode.log
  tail -3 $O/probe_$mode.log
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
