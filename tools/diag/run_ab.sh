set -e
cd $GRAFT_REPO_ROOT
export MKGNN_NO_SMALL_BATCH=1
for a in 2 0 4; do echo "ahead $a"; MKGNN_SHARD_AHEAD=$a python3 bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline 2>&1 | grep -o "shard epoch: [^,]*,[^,]*\|Error.*"; done
