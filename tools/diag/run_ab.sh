set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "readout or head or network" > gpurun_out/t7.log 2>&1 || { tail -30 gpurun_out/t7.log; exit 1; }
tail -2 gpurun_out/t7.log
MKGNN_NO_SHARD_EPOCH=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/b7.json 2> gpurun_out/b7.err
grep -o '"ms_per_step": [0-9.]*' gpurun_out/b7.json
export MKGNN_NO_SMALL_BATCH=1
timeout -k 10 200 bash tools/prof.sh r03_prof7 bench.py --steps 20 --warmup 5 --windows 1 --fresh-batches 0 --no-cpu-baseline > gpurun_out/prof7.txt
python3 tools/step_timeline.py gpurun_out/r03_prof7 > gpurun_out/prof7_timeline.txt
find gpurun_out/r03_prof7 -name "*kernel_trace.csv" -delete
