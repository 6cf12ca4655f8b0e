#!/bin/bash
O=gpurun_out/r02v; mkdir -p $O
timeout 600 python -m pytest tests/test_dense_gpu.py -x -q -m gpu 2>&1 | tail -3
for f in "" "-DPTT_SAS_INTERLEAVE=0"; do
  PTT_MFMA_FLAGS="$f" python -m ptt_amd.build --force > $O/build.log 2>&1
  echo "== flags [$f]"; timeout 200 python scripts/kernel_bench.py --only sa1_s,sa2_s --iters 50 2>&1 | grep hoist
done
python -m ptt_amd.build --force > $O/build.log 2>&1
