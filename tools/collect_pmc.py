"""Aggregate rocprofv3 --pmc passes (tools/pmc.sh: one directory per pass) of one kernel into the JSON bench.py reads for
roofline.traffic.  usage: tools/collect_pmc.py <kernel substring> <out.json> <commit> <algorithmic bytes> <dir with pass*/>"""
import csv
import glob
import json
import os
import sys

kernel, out, commit, alg_bytes, root = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
workload = sys.argv[6] if len(sys.argv) > 6 else "tools/fwd_probe.py (batch 4096 molecules, ~102.5 k atoms, N-hop layer F=110, training configuration)"
sums, counts, name = {}, {}, None
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = r["Counter_Name"]
        sums[k] = sums.get(k, 0.0) + float(r["Counter_Value"])
        counts.setdefault(k, set()).add((f, r["Dispatch_Id"]))
avg = {k: sums[k] / len(counts[k]) for k in sums}
fetch_kb, write_kb = avg["FETCH_SIZE"], avg["WRITE_SIZE"]
hit, miss = avg.get("TCC_HIT_sum"), avg.get("TCC_MISS_sum")
res = {
    "kernel": name,
    "commit": commit,
    "kernel_source_sha16": __import__("hashlib").sha256(b"".join(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "molkgnn_amd", "csrc", f), "rb").read() for f in ("kgnn_fwd_stream.hip", "kgnn_split.h"))).hexdigest()[:16],
    "workload": workload,
    "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum, separate passes "
                 f"(tools/pmc.sh), averages over {len(counts['FETCH_SIZE'])} launches",
    "FETCH_SIZE_KB_raw": fetch_kb,
    "WRITE_SIZE_KB": write_kb,
    "TCC_HIT_sum": hit,
    "TCC_MISS_sum": miss,
    "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss is not None else None,
    "correction": "gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes for 16-byte-per-lane loads (MI355X_MICROARCH.md, "
                  "HBM section; the row gather of this kernel is 16-byte-per-lane LDS-DMA): read bytes = 2 x FETCH_SIZE x 1024; "
                  "WRITE_SIZE x 1024 taken as is (4-byte-per-lane stores are uncalibrated in the guide)",
    "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
    "hbm_bytes_per_launch_uncorrected": int(fetch_kb * 1024 + write_kb * 1024),
    "algorithmic_bytes_per_launch": alg_bytes,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
