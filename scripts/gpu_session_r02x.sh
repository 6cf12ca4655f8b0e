#!/bin/bash
O=gpurun_out/r02x; mkdir -p $O
for e in 0 1 2 4 8 15; do
  PTT_MFMA_FLAGS="-DPTT_SAL_EXP=$e" python -m ptt_amd.build --force > $O/build.log 2>&1
  echo "== EXP $e"; timeout 200 python scripts/kernel_bench.py --only sa0_s --iters 50 2>&1 | grep sa0
done
python -m ptt_amd.build --force > $O/build.log 2>&1
