#!/bin/bash
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_hot_path_gpu.py -x -q -m gpu 2>&1 | tail -1
timeout 300 python scripts/kernel_bench.py --only pair,sa0,sa1,sa2,sa_box,xcorr --iters 40 2>&1 | grep -v amdgpu | grep -v "^sa[12]_s  "
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --sustain 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['kernel_ms_per_step'], d['full_model']['ms_per_step'])"; done
