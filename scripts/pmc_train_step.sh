#!/bin/bash
# HBM-side traffic of ONE training step (BASELINE.json configs[3]): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in two separate
# counter-only runs of `bench.py --workload train --no-extras` with PTT_TRAIN_GRAPH=0 (the eager step: the same launches the captured
# step replays, and the step count of the run is exactly steps + warm-up), every dispatch summed, divided by the
# steps the run executed -> <out>/pmc_summary.json in the form bench.committed_traffic() reads ("whole training step").
#   bash scripts/pmc_train_step.sh gpurun_out/r05p/pmc_train_step
set -u
OUT=${1:?output directory}
STEPS=4; WARM=1
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_ts_$c
    PTT_TRAIN_GRAPH=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_ts_$c -- python "$REPO/bench.py" --workload train --steps $STEPS --warmup $WARM --no-extras \
        --sustain 0 --no-cpu-baseline > /tmp/pmc_ts_$c.log 2>&1
    f=$(find /tmp/pmc_ts_$c -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$REPO/$OUT/pmc_$c.csv" || tail -5 /tmp/pmc_ts_$c.log > "$REPO/$OUT/pmc_$c.err"
done
cd "$REPO" && python - "$OUT" $((STEPS + WARM)) <<'PY'
import csv, json, os, sys
from collections import defaultdict
d, steps = sys.argv[1], int(sys.argv[2])
tot, per_kernel, launches = {}, defaultdict(lambda: defaultdict(float)), 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    s = 0.0
    n = 0
    with open(os.path.join(d, "pmc_%s.csv" % c)) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != c:
                continue
            v = float(r["Counter_Value"])
            s += v
            n += 1
            per_kernel[r["Kernel_Name"].split("(")[0].replace("void ", "")][c] += v
    tot[c], launches = s, max(launches, n)
read_b, write_b = 2.0 * tot["FETCH_SIZE"] * 1024 / steps, tot["WRITE_SIZE"] * 1024 / steps        # gfx950: 128-B requests tallied at 64 B
top = sorted(per_kernel.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:12]
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two counter-only runs of bench.py --workload train (%d steps each, every "
               "dispatch summed, divided by the steps); read bytes = 2 x FETCH_SIZE KiB x 1024 on gfx950, write = WRITE_SIZE KiB x 1024" % steps,
       "kernels": {"whole training step": {"read_bytes": read_b, "write_bytes": write_b, "launches_per_step": launches / steps,
                                           "steps_profiled": steps}},
       "largest_kernels_bytes_per_step": {k: {"read_bytes": 2.0 * v["FETCH_SIZE"] * 1024 / steps, "write_bytes": v["WRITE_SIZE"] * 1024 / steps} for k, v in top}}
try:
    out["build"] = open(os.path.join("ptt_amd", "lib", "BUILD_ID")).read().strip()
except OSError:
    out["build"] = "unknown"
json.dump(out, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)
print("training step: %.2f GB read + %.2f GB written per step, %.0f launches" % (read_b / 1e9, write_b / 1e9, launches / steps))
PY
