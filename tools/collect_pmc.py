#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (one directory per pass) for one kernel into the JSON that bench.py reads.
usage: tools/collect_pmc.py <kernel substring> <out.json> <pass_dir> [<pass_dir> ...]"""
import csv
import glob
import json
import os
import sys

kernel, out = sys.argv[1], sys.argv[2]
sums, counts = {}, {}
for d in sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel not in r["Kernel_Name"]:
                continue
            k = r["Counter_Name"]
            sums[k] = sums.get(k, 0.0) + float(r["Counter_Value"])
            counts.setdefault(k, set()).add(r["Dispatch_Id"])
avg = {k: sums[k] / len(counts[k]) for k in sums}
print({k: (avg[k], len(counts[k])) for k in avg})
fetch_kb, write_kb = avg["FETCH_SIZE"], avg["WRITE_SIZE"]
hit, miss = avg.get("TCC_HIT_sum"), avg.get("TCC_MISS_sum")
res = {
    "kernel": "kc_forward_fused<7>",
    "workload": "tools/fwd_probe.py --reps 5 (batch 4096 molecules, 102583 atoms, N-hop layer F=110, training configuration)",
    "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, three separate passes, "
                 f"averages over {len(counts['FETCH_SIZE'])} launches",
    "FETCH_SIZE_KB_raw": fetch_kb,
    "WRITE_SIZE_KB": write_kb,
    "TCC_HIT_sum": hit,
    "TCC_MISS_sum": miss,
    "l2_hit_rate": (hit / (hit + miss)) if hit is not None else None,
    "correction": "gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes for 16-byte-per-lane loads (MI355X_MICROARCH.md, "
                  "HBM section): read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 taken as is",
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "hbm_bytes_per_launch_uncorrected": int(fetch_kb * 1024 + write_kb * 1024),
    "algorithmic_bytes_per_launch": 98774728,
    "note": "traffic above the algorithmic bytes = the gather reads every row once per role (focal + each neighbour role: ~3.2x the "
            "x rows), degree 3/4 gather once per column part, and the training configuration writes permutation ids and three score planes",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
