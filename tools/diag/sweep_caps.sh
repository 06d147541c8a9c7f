# sweep of the backward's grid caps (bank x rows), bench.py ms_per_step: tools/diag/sweep_caps.sh
export MKGNN_NO_SMALL_BATCH=1 MKGNN_NO_SHARD_EPOCH=1
for bank in 256 320 384 448; do for rows in 384 448 512; do
  v=$(MKGNN_BANK_STREAM_BLOCKS=$bank MKGNN_ROWS_STREAM_BLOCKS=$rows python3 bench.py --steps 30 --warmup 5 --fresh-batches 0 --no-cpu-baseline --roofline-reps 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "bank $bank rows $rows $v"
done; done
