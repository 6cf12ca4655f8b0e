#!/bin/bash
# sa_stream_kernel (in-stream pool form): tiles per workgroup, standalone and inside the graphed step
O=gpurun_out/r03d; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
for c in 2 3 4 6 12; do
  echo "== PTT_SA_CHUNK=$c"
  PTT_SA_CHUNK=$c timeout 200 python scripts/kernel_bench.py --only sa1_s,sa2_s --iters 50 2>&1 | grep hoist
  PTT_SA_CHUNK=$c timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-full-model --sustain 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   bench', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['kernel_ms_per_step']['sa_fused_fwd'])"
done
python -m ptt_amd.build --force > $O/build.log 2>&1
