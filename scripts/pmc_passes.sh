#!/bin/bash
# rocprofv3 PMC passes (counters only: no --kernel-trace / --stats / trace domains in the same run) on
# scripts/kernel_bench.py for the dense kernels of the hot path, then a per-kernel summary.
#   bash scripts/pmc_passes.sh gpurun_out/r02_pmc ["pair,sa0_s,sa1_s,sa2_s,sa_box"]
# Each pass is a separate process (8 SQ slots; FETCH_SIZE / WRITE_SIZE do not fit one TCC pass).
set -u
OUT=${1:-gpurun_out/pmc}
ONLY=${2:-pair,sa0_s,sa1_s,sa2_s,sa_box}
# optional third argument: another command to profile instead of scripts/kernel_bench.py (e.g. the training row GEMMs:
#   bash scripts/pmc_passes.sh gpurun_out/x_pmc - "python scripts/rows_gemm_bench.py --no-check --pmc")
CMD=${3:-}
# optional fourth argument: extra arguments for scripts/kernel_bench.py (e.g. "--batch 32 --pair-n 2048,64": the stress shapes)
KB_ARGS=${4:-}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
pass() {
    name=$1; shift
    rm -rf /tmp/pmc_$name
    if [ -n "$CMD" ]; then
        (cd "$REPO" && timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- $CMD > /tmp/pmc_$name.log 2>&1)
    else
        timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- \
            python "$REPO/scripts/kernel_bench.py" --only "$ONLY" --iters 4 $KB_ARGS > /tmp/pmc_$name.log 2>&1
    fi
    f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
        grep -E "Counter_Name|pt_attn_pair|sa_fused_kernel|sa_wave_kernel|sa_stream_kernel|sa_lds_kernel|linear_kernel|linear_small_kernel|xcorr_fused|rows_gemm_kernel|wgrad2_kernel|linear_wgrad_kernel|rowjobs_kernel|wgrad_stream_kernel|sa_z0_bnbwd_kernel|xcorr_z0_bnbwd_kernel|scatter_csr_count_kernel" "$f" > "$REPO/$OUT/pmc_$name.csv"
    else
        tail -5 /tmp/pmc_$name.log > "$REPO/$OUT/pmc_$name.err"
    fi
}
pass mfma   SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES
pass issue  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass grbm   GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass fetch  FETCH_SIZE
pass write  WRITE_SIZE
pass tcc    TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
cd "$REPO" && python scripts/pmc_summary.py "$OUT"
