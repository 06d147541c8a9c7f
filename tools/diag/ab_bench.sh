# A/B of library builds on the headline step: tools/diag/ab_bench.sh <variant> [<variant> ...]   ("main" = the tree's); two rounds
export MKGNN_NO_SMALL_BATCH=1 MKGNN_NO_SHARD_EPOCH=1
for i in 1 2; do for v in "$@"; do
  if [ "$v" = main ]; then unset MKGNN_LIB; else export MKGNN_LIB=build_variants/$v/libmolkgnn_hip.so; fi
  printf "%-12s " "$v"; python3 bench.py --steps 30 --warmup 5 --fresh-batches 0 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], [(k['kernel'][:28], k['ms_per_launch']) for k in d['roofline']['kernels']])"
done; done
