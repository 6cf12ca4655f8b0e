#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_step_ops_gpu.py tests/test_round5_gpu.py -q -m gpu -s 2>&1 | grep -E "ClipAdam vs|fused BatchNorm|passed|failed|FAILED|Error" | cut -c1-250
