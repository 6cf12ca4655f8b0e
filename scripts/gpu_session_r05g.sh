#!/bin/bash
# Round 5, session g: the whole suite without -x (which tests the fused BatchNorm backward breaks), wgrad tile shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05g; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "^FAILED|passed|failed" $O/pytest_gpu.log | cut -c1-200
PTT_FUSED_BN_BWD=0 timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_train_config3_gpu.py -q -m gpu > $O/pytest_unfused.log 2>&1; grep -E "^FAILED|passed|failed" $O/pytest_unfused.log | cut -c1-200
WG_FIRSTS=0,1,2,3 timeout 600 python scripts/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_bench.log
