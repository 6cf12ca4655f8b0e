#!/bin/bash
set -u
O=gpurun_out/r02e
mkdir -p $O
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_dense_gpu.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-full-model > $O/bench_car.json 2> $O/bench_car.err; python - <<PY
import json
d=json.loads(open("$O/bench_car.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["latency_b1"])
PY
tail -3 $O/bench_car.err
