#!/bin/bash
# Copies the merged output of a closing GPU session (gpurun_out/r<NN><S>, scripts/gpu_sessions_r<NN>.sh z / p) into profiles/r<NN><S>_*.
#   bash scripts/copy_closing_evidence.sh z [06]
set -e
S=${1:-z}
R=${2:-05}
O=gpurun_out/r$R$S
for f in pytest_gpu.log bench_default.json bench_train.json serial_kernel_stats.csv serial_bench_line.json train_kernel_stats.csv \
         train_profiled_bench_line.json b1_kernel_stats.csv sa_z0_bnbwd_probe.log train_stream_kernels_bench.log; do
    [ -f $O/$f ] && cp $O/$f profiles/r$R${S}_$f
done
for f in last_step_by_shape.txt last_step_launches.txt; do
    [ -f $O/train_timeline/$f ] && cp $O/train_timeline/$f profiles/r$R${S}_train_$f
done
for d in pmc pmc_stress pmc_stress_sampling_order pmc_train_gemm pmc_train_stream pmc_train_step; do
    if [ -f $O/$d/pmc_summary.json ]; then mkdir -p profiles/r$R${S}_$d; cp $O/$d/*.csv $O/$d/pmc_summary.json profiles/r$R${S}_$d/; fi
done
