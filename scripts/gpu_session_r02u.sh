#!/bin/bash
# where does sa_stream_kernel's time go: timing-only variants (results wrong) with phases compiled out
O=gpurun_out/r02u; mkdir -p $O
for e in 0 1 2 4 8 16 3 31; do
  PTT_MFMA_FLAGS="-DPTT_SAS_EXP=$e" python -m ptt_amd.build --force > $O/build_$e.log 2>&1
  echo "== EXP $e"; timeout 200 python scripts/kernel_bench.py --only sa1_s,sa2_s --iters 50 2>&1 | grep hoist
done
