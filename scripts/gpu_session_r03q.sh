#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
for T in 0 256 128 1024; do echo "== PTT_FPS_T=$T"; PTT_FPS_T=$T timeout 300 python scripts/fps_vs_pair_probe.py 2>&1 | grep -v amdgpu; done
python -m ptt_amd.build --force > $O/build.log 2>&1
