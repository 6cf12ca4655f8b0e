#!/bin/bash
set -u
O=gpurun_out/r02l
mkdir -p $O
timeout 600 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -2
timeout 600 python bench.py --workload train --steps 10 --warmup 3 --sustain 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('release', d['value'], d['ms_per_step'])"
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
for tile in 11 12 21 22; do
PTT_LINEAR_TILE=$tile timeout 600 python bench.py --workload train --steps 10 --warmup 3 --sustain 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $tile', d['value'], d['ms_per_step'])"
done
