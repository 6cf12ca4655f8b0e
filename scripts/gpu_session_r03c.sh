#!/bin/bash
# ball query with four centres per wave: parity, then the stress and car lines
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_point_ops_gpu.py tests/test_hot_path_gpu.py -x -q -m gpu 2>&1 | tail -3
for w in stress car; do
timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-full-model > $O/bench_$w.json 2> $O/bench_$w.err
python -c "
import json
d = json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1])
print('$w', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['index_ops']['ball_query'])"
done
