#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
timeout 300 python scripts/fps_sweep.py 2>&1 | grep -v amdgpu
python -m ptt_amd.build --force > $O/build.log 2>&1
