#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O
for f in "" "-DPTT_PAIR_PF=0" "-DPTT_PAIR_PF=2"; do
  PTT_MFMA_FLAGS="$f" python -m ptt_amd.build --force > $O/build.log 2>&1
  echo "== flags [$f]"; timeout 200 python scripts/kernel_bench.py --only pair --iters 40 2>&1 | grep pair
done
python -m ptt_amd.build --force > $O/build.log 2>&1
