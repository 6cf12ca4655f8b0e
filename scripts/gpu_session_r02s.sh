#!/bin/bash
# full tracker at B=48: per-kernel device time of one eager forward (what the 1.08 ms beyond the hot path is made of)
O=gpurun_out/r02s; mkdir -p $O
timeout 400 python scripts/full_model_profile.py 2>&1 | grep -v amdgpu | tee $O/full_model_profile.txt
