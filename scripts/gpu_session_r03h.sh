#!/bin/bash
# FPS footprint vs the pipelined step
O=gpurun_out/r03h; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
for T in 0 1024 0 1024; do
  echo "== PTT_FPS_T=$T"
  PTT_FPS_T=$T timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-full-model --sustain 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   bench', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['kernel_ms_per_step']['fps'])"
done
python -m ptt_amd.build --force > $O/build.log 2>&1
