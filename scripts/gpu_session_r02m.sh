#!/bin/bash
set -u
O=gpurun_out/r02m
mkdir -p $O
timeout 1200 python -m pytest tests/test_dense_gpu.py tests/test_hot_path_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -3
timeout 300 python scripts/kernel_bench.py --only sa0 --iters 20 2>&1 | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline --no-latency --no-full-model > $O/bench_car.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench_car.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["sustained"], d["kernel_ms_per_step"])
PY
