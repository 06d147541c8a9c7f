"""Print the fields of a bench.py line that a round's notes quote: tools/diag/bench_summary.py <file with the JSON line>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "graph_replay", d.get("graph_replay"), "n_gpus", d["n_gpus"])
r = d["roofline"]
print("roofline frac", r["frac"], "ms", r["ms_per_launch"], "traffic", r["traffic"], "|", r["traffic_source"])
for k in r["kernels"]:
    print("   ", k["kernel"], k["ms_per_launch"], "hbm", k["hbm_frac"], "fp32", k["fp32_frac"], "traffic", k.get("traffic"))
fb = d.get("fresh_batches", {})
print("fresh", fb.get("ms_per_step"), "shards", fb.get("shard_epoch", {}).get("ms_per_step"))
for k, v in d.get("small_batch", {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print("small", k, v["ms_per_step"], "fwd kernel ms", v["forward_kernel_ms"], "traffic", v.get("forward_kernel_traffic"),
              "eager", v["paths"]["eager_ms_per_step"]["per_operator"], v["paths"]["eager_ms_per_step"]["molecule_resident"])
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
