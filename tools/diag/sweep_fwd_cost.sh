#!/bin/bash
# Forward block split: per-degree cost candidates (MKGNN_STREAM_COST="c1,c2,c3,c4", units of 32 cycles per tile) against the
# launch time of tools/fwd_probe.py:  tools/diag/sweep_fwd_cost.sh <out file> [fwd_probe arguments]
out="$1"; shift
: > "$out"
for c in "" "319,475,853,1056" "196,337,533,710" "228,404,596,811" "210,337,533,760" "230,337,533,710" "196,360,533,710" "196,337,570,710" "196,337,533,780" "131,176,303,462" "150,176,303,462" "131,190,303,462" "131,176,330,462" "131,176,303,420"; do
    r=$(MKGNN_STREAM_COST="$c" python3 tools/fwd_probe.py --reps 40 "$@" 2>/dev/null | grep "launch us")
    echo "cost [$c] $r" >> "$out"
done
cat "$out"
