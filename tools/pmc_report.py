"""Per-kernel averages of the rocprofv3 --pmc passes under a directory: tools/pmc_report.py <dir> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "mkgnn"
sums = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
for k in sorted(sums):
    print(k)
    for c in sorted(sums[k]):
        n = len(disp[k][c])
        print(f"    {c:32s} {sums[k][c] / n:16.1f}   ({n} launches)")
