#!/bin/bash
set -u
O=gpurun_out/r02k
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python - <<PY
import json
d=json.loads(open("$O/bench_train.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["sustained"], d["loss"])
PY
