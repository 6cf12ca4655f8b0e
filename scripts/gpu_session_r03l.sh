#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_car.json 2> $O/bench_car.err
python -c "
import json
d = json.loads(open('$O/bench_car.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'full', d['full_model']['value'], d['full_model']['ms_per_step'], d['latency_b1']['full_tracker_ms_per_frame'], d['latency_b1']['tracklet_loop']['b1'], d['latency_b1']['tracklet_loop']['b48'])"
