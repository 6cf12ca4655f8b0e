#!/bin/bash
# stagger quanta of the co-resident workgroups, measured inside the graphed step
O=gpurun_out/r03v; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-full-model --sustain 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ', d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['kernel_ms_per_step']['pt_attn_pair'], d['kernel_ms_per_step']['sa_fused_fwd'])"; }
for rep in 1 2 3; do
  echo "== default (pair 8, sa 2)"; run PTT_PAIR_STAGGER=8 PTT_SA_STAGGER=2
  echo "== none (pair 0, sa 0)"; run PTT_PAIR_STAGGER=0 PTT_SA_STAGGER=0
done
python -m ptt_amd.build --force > $O/build.log 2>&1
