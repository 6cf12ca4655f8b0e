#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
for f in "" "-DPTT_LINEAR_PF=0" "" "-DPTT_LINEAR_PF=0"; do
  PTT_MFMA_FLAGS="$f" python -m ptt_amd.build --force > $O/build.log 2>&1
  echo "== flags [$f]"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --sustain 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   bench', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['kernel_ms_per_step']['linear'], 'full', d['full_model']['ms_per_step'])"
done
python -m ptt_amd.build --force > $O/build.log 2>&1
