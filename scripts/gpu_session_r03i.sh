#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
for c in 1 2 3 2 1 3; do
  echo "== PTT_SA_LDS_CHUNK=$c"
  PTT_SA_LDS_CHUNK=$c timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-full-model --sustain 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   bench', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['kernel_ms_per_step']['sa_fused_fwd'])"
done
python -m ptt_amd.build --force > $O/build.log 2>&1
