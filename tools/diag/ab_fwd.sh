# A/B of forward-kernel builds: tools/diag/ab_fwd.sh <width> <variant> [<variant> ...]   (build_variants/<variant>/libmolkgnn_hip.so; "main" = the tree's)
w=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset MKGNN_LIB; else export MKGNN_LIB=build_variants/$v/libmolkgnn_hip.so; fi
  printf "%-22s " "$v"; timeout -k 5 120 python tools/fwd_probe.py --width $w 2>&1 | grep "fused forward" || echo failed
done
