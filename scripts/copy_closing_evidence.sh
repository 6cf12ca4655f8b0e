#!/bin/bash
# Copies the merged output of the closing GPU session (gpurun_out/r03z, scripts/gpu_sessions_r03.sh z) into profiles/r03z_*.
set -e
O=gpurun_out/r03z
for f in pytest_gpu.log bench_default.json bench_train.json serial_kernel_stats.csv serial_bench_line.json train_kernel_stats.csv \
         train_profiled_bench_line.json b1_kernel_stats.csv forward_precision_vs_float64.log gradient_error_vs_float64.log; do
    cp $O/$f profiles/r03z_$f
done
cp $O/pmc/*.csv $O/pmc/pmc_summary.json profiles/r03z_pmc/
cp $O/pmc_train_gemm/*.csv $O/pmc_train_gemm/pmc_summary.json profiles/r03z_pmc_train_gemm/
