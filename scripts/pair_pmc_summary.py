#!/usr/bin/env python
"""Derive profiles/r01_pair_kernel_pmc.json's fields from the three rocprofv3 --pmc CSVs (scripts/pair_pmc.sh)."""
import csv, json, sys
from collections import defaultdict

d = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))            # grid -> counter -> values per launch
for name in ("fetch", "write", "tcc"):
    with open("%s/pair_pmc_%s.csv" % (d, name)) as f:
        for r in csv.DictReader(f):
            vals[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc passes (separate runs, no trace domains) on scripts/kernel_bench.py --only pair, B=48; "
               "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B "
               "(MI355X_MICROARCH.md HBM section) so read bytes = 2 x FETCH_SIZE x 1024. Infinity-Cache hits are "
               "counted at this interface, so this is fabric traffic, an upper bound on HBM traffic.",
       "kernel": "pt_attn_pair_kernel<512>", "launches": {}}
for grid, c in sorted(vals.items(), reverse=True):
    N = grid // 256 * 2 // 48                               # 256 threads per workgroup, 2 points per workgroup, B = 48
    mean = lambda k: sum(c[k]) / max(1, len(c[k]))
    rd, wr = 2.0 * mean("FETCH_SIZE") * 1024, mean("WRITE_SIZE") * 1024
    hit, miss = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
    rows = 48 * N
    # compulsory bytes: q|k|v rows, neighbour indices, relative coordinates, result rows, the three 512x512 weights
    alg = rows * (1536 * 4 + 16 * 4 + 16 * 12 + 512 * 4) + 3 * 512 * 512 * 4 + 6 * 512 * 4
    out["launches"]["B48_N%d" % N] = {
        "FETCH_SIZE_KiB": mean("FETCH_SIZE"), "WRITE_SIZE_KiB": mean("WRITE_SIZE"), "read_bytes_corrected": rd,
        "write_bytes": wr, "traffic_bytes": rd + wr, "TCC_HIT": hit, "TCC_MISS": miss,
        "l2_hit_rate": hit / max(1.0, hit + miss), "algorithmic_bytes": alg}
json.dump(out, open("%s/pair_kernel_pmc.json" % d, "w"), indent=1)
print(json.dumps(out["launches"], indent=1))
