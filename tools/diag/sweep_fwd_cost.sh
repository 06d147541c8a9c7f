#!/bin/bash
# Forward block split: per-degree cost candidates (MKGNN_STREAM_COST="c1,c2,c3,c4", units of 32 cycles per tile) against the
# launch time of tools/fwd_probe.py and the kernel span of tools/stream_stamps.py:
#   tools/diag/sweep_fwd_cost.sh <out file> "<c1,c2,c3,c4>" ... [-- fwd_probe arguments]
out="$1"; shift
cands=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do cands+=("$1"); shift; done
[ "$1" = "--" ] && shift
: > "$out"
for c in "${cands[@]}"; do
    r=$(MKGNN_STREAM_COST="$c" python3 tools/fwd_probe.py --reps 100 "$@" 2>/dev/null | grep "launch us")
    s=""
    for i in 1 2 3; do s="$s $(MKGNN_STREAM_COST="$c" python3 tools/stream_stamps.py "$@" 2>/dev/null | grep -o 'kernel span [0-9.]*' | cut -d' ' -f3)"; done
    echo "cost [$c] $r | spans$s" >> "$out"
done
cat "$out"
