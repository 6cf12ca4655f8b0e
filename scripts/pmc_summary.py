#!/usr/bin/env python
"""Per-kernel summary of the rocprofv3 --pmc CSVs written by scripts/pmc_passes.sh -> <dir>/pmc_summary.json.

For every (kernel, grid size) the mean of each counter over the profiled launches, plus derived figures:
  mfma_busy_frac   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of SIMD-cycles (of busy CUs) in which
                   the matrix pipe was executing (MI355X_MICROARCH.md: the counter ticks in cycles, summed over SIMDs)
  mfma_flops       SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 (one MOP = 512 FLOP on gfx94x/gfx950 derived-counter tables)
  read/write bytes 2 x FETCH_SIZE KiB (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE KiB
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))            # (kernel, grid) -> counter -> values per launch
for path in sorted(glob.glob(os.path.join(d, "pmc_*.csv"))):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ptt::", "")
            vals[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc passes (separate processes, counters only) on scripts/kernel_bench.py, B = 48; means "
               "over the profiled launches of each (kernel, grid). FETCH_SIZE / WRITE_SIZE in KiB; read bytes = 2 x "
               "FETCH_SIZE x 1024 on gfx950.", "kernels": {}}
for (k, grid), c in sorted(vals.items()):
    m = {name: sum(v) / len(v) for name, v in c.items()}
    e = {"grid_threads": grid, "workgroups": grid // 256, "counters": m, "launches_profiled": max(len(v) for v in c.values())}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_BUSY_CU_CYCLES"])
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
        e["mfma_flops"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
    if "FETCH_SIZE" in m:
        e["read_bytes"] = 2.0 * m["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in m:
        e["write_bytes"] = m["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in m:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0.0))
    if m.get("SQ_WAVE_CYCLES"):
        for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if w in m:
                e[w.lower() + "_frac_of_wave_cycles"] = m[w] / m["SQ_WAVE_CYCLES"]
    out["kernels"]["%s grid=%d" % (k, grid)] = e
try:
    out["build"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ptt_amd", "lib", "BUILD_ID")).read().strip()
except OSError:
    out["build"] = "unknown (no ptt_amd/lib/BUILD_ID: see the directory name for the session)"
json.dump(out, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in e.items() if x != "counters"})
