#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
timeout 1500 python -m pytest tests/test_point_ops_gpu.py tests/test_golden_gpu.py tests/test_hot_path_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 400 python bench.py --workload stress --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stress', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
