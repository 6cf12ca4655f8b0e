#!/bin/bash
set -u
O=gpurun_out/r02n
mkdir -p $O
PTT_HIP_FLAGS=-DPTT_DEV python -m ptt_amd.build --force > $O/build.log 2>&1
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-latency --no-full-model --sustain 1 $EXTRA > $O/b_$tag.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["value"], d["ms_per_step"], d["sustained"]["ms_per_step"], d["kernel_ms_per_step"]["sa_fused_fwd"])
PY
}
for rep in 1 2; do
for mode in "" "--no-pipeline" "--serial"; do
  EXTRA="$mode"
  run wave$mode$rep PTT_SA_LDS=0
  run lds$mode$rep PTT_SA_LDS=1
done; done
