#!/bin/bash
# sa_stream v2 (pooling / meta inside the layer-1 MFMA stream, two barriers per tile): parity, then the bench line
O=gpurun_out/r02t; mkdir -p $O
timeout 900 python -m pytest tests/test_dense_gpu.py tests/test_hot_path_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $O/bench_car.json 2> $O/bench_car.err; python -c "
import sys, json
d = json.loads(open('gpurun_out/r02t/bench_car.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'], d.get('sustained'))"
