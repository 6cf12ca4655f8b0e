#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
SWEEP_LINEAR=1 timeout 300 python scripts/kernel_bench.py --only lin --iters 100 2>&1 | grep -v amdgpu
python -m ptt_amd.build --force > $O/build.log 2>&1
