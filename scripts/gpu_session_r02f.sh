#!/bin/bash
set -u
O=gpurun_out/r02f
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/bench_car.json 2> $O/bench_car.err; python - <<PY
import json
d=json.loads(open("$O/bench_car.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["full_model"]["value"], json.dumps(d["latency_b1"]))
PY
tail -3 $O/bench_car.err
