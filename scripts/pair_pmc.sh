#!/bin/bash
# rocprofv3 PMC passes for the dominant kernel (separate runs, counters only — no trace domains), then the
# per-launch summary bench.py's roofline.traffic reads. Run on the GPU box from the repo root:
#   bash scripts/pair_pmc.sh gpurun_out/pmc
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/pmc_$name
    rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$name -- python "$REPO/scripts/kernel_bench.py" --only pair --iters 4 > /dev/null 2>&1
    f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|pt_attn_pair" "$f" > "$REPO/$OUT/pair_pmc_$name.csv"
done
cd "$REPO" && python scripts/pair_pmc_summary.py "$OUT"
