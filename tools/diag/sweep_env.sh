#!/bin/bash
# bench.py ms_per_step under a list of environment settings:  tools/diag/sweep_env.sh "A=1 B=2" "C=3" ...   ("" = defaults)
export MKGNN_NO_SMALL_BATCH=1 MKGNN_NO_SHARD_EPOCH=1
for setting in "$@"; do
  v=$(env $setting python3 bench.py --steps 40 --warmup 5 --fresh-batches 0 --no-cpu-baseline --roofline-reps 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "[$setting] $v"
done
