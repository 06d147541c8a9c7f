"""One training step's kernels, in launch order, from a rocprofv3 --kernel-trace CSV of bench.py:
tools/step_timeline.py <dir or kernel_trace.csv> > profiles/...txt   (the step is delimited by the batch norm's statistics launches)"""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("bn_forward_fused_kernel", "bn_stats_kernel", "bn_stats_prep_kernel", "bn_sum_kernel"))]
s, e = marks[-3], marks[-2]
t0 = int(rows[s]["Start_Timestamp"])
print("# start_us  duration_us  queue  kernel")
total = 0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    total += en - st
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"at::native::.*?(\w+)<.*", r"at::native::\1", name)
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {name[:90]}")
print(f"# kernels {e - s}, sum of kernel durations {total / 1e3:.1f} us, span of the step {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us")
