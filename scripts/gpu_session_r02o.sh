#!/bin/bash
mkdir -p gpurun_out/r02o
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > gpurun_out/r02o/build.log 2>&1
timeout 300 python scripts/fps_sweep.py 2>&1 | grep -v amdgpu
