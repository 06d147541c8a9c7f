#!/bin/bash
# rocprofv3 kernel statistics of one command:  tools/prof.sh <out dir under gpurun_out/> <python script> [args ...]
# (run from the repository root on the GPU box; the program goes straight after `--`, no wrapper in between)
set -e
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
script="$GRAFT_REPO_ROOT/$1"; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$script" "$@" > "$out/run.log" 2>&1
cd "$GRAFT_REPO_ROOT" && python3 tools/kstats.py "$out" 12
