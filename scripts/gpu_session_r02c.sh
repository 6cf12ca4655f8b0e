#!/bin/bash
set -u
O=gpurun_out/r02c
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 300 python scripts/kernel_bench.py --only sa,pair --iters 20 > $O/kernel_bench.log 2>&1; cat $O/kernel_bench.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_car.json 2> $O/bench_car.err; python - <<PY
import json
d=json.loads(open("$O/bench_car.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["sustained"], d["kernel_ms_per_step"], d["full_model"], d["latency_b1"])
PY
tail -5 $O/bench_car.err
