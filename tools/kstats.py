"""Print the top rows of a rocprofv3 kernel_stats.csv: tools/kstats.py <dir or csv> [rows]."""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(f"{path}: total {sum(float(r['TotalDurationNs']) for r in rows) / 1e6:.2f} ms of kernel time")
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.1f} %")
