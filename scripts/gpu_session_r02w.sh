#!/bin/bash
O=gpurun_out/r02w; mkdir -p $O
timeout 600 python -m pytest tests/test_dense_gpu.py -x -q -m gpu 2>&1 | tail -2
for f in "-DPTT_SAL_WAVES=8 -DPTT_SAL_WGS=1" "-DPTT_SAL_WAVES=8 -DPTT_SAL_WGS=2" "-DPTT_SAL_WAVES=12 -DPTT_SAL_WGS=1"; do
  PTT_MFMA_FLAGS="$f" python -m ptt_amd.build --force > $O/build.log 2>&1
  echo "== flags [$f]"; timeout 200 python scripts/kernel_bench.py --only sa0_s --iters 50 2>&1 | grep sa0
  timeout 200 python scripts/kernel_bench.py --only sa0_s --iters 50 --batch 24 2>&1 | grep sa0
done
python -m ptt_amd.build --force > $O/build.log 2>&1
