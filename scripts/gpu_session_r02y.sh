#!/bin/bash
# vote-aggregation level (sa_fused_kernel<16,RT>): 64-row vs 32-row workgroups (768 vs 1536 workgroups on 512 slots)
O=gpurun_out/r02y; mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
for rt in 2 1; do echo "== PTT_SA_RT=$rt"; PTT_SA_RT=$rt timeout 200 python scripts/kernel_bench.py --only sa_box --iters 50 2>&1 | grep sa_box; done
python -m ptt_amd.build --force > $O/build.log 2>&1
