#!/bin/bash
# rocprofv3 PMC passes of one command, one pass per counter group (never combined with a trace domain other than
# --kernel-trace):  tools/pmc.sh <out dir under gpurun_out/> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <python script> [args ...]
# prints per-kernel averages per counter (tools/pmc_report.py).  Run from the repository root on the GPU box.
set -e
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
script="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
    rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/pass$i" -- python3 "$script" "$@" > "$out/pass$i.log" 2>&1 || { tail -5 "$out/pass$i.log"; }
    i=$((i + 1))
done
cd "$GRAFT_REPO_ROOT" && python3 tools/pmc_report.py "$out"
