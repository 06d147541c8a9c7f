#!/bin/bash
# Every measured artefact of a round in one GPU call, written under gpurun_out/<tag>/ (copy what is to be judged into
# profiles/ afterwards: tools/round_profiles_collect.sh <tag>):  tools/round_profiles.sh <tag> <commit> [pmc|rest]
# ("pmc": only the forward kernels' HBM-traffic passes -- run them first, copy the three forward_pmc*.json into profiles/ and
# commit, so that the bench lines of the "rest" call quote traffic measured on the same kernels)
# Run from the repository root on the GPU box.  Programs go straight after `--` under rocprofv3 (no wrappers); the PMC
# passes carry --kernel-trace only.
set -e
tag="$1"; commit="$2"; part="${3:-all}"
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/$tag"
mkdir -p "$O"
cd "$R"
if [ "$part" != pmc ]; then
# 1. the bench line (configs[1], batch 4096) with roofline, fresh-batch leg and CPU baseline
python3 bench.py --steps 50 --warmup 10 > "$O/bench_b4096.json" 2> "$O/bench_b4096.err"
# 2. kernel statistics + one step's timeline of the same command (shorter; without the small-batch leg, whose steps would
#    otherwise be the last ones of the trace)
export MKGNN_NO_SMALL_BATCH=1 MKGNN_NO_EXACT_LEG=1       # (the traced steps are the headline's: no leg behind them)
tools/prof.sh "$tag/prof_b4096" bench.py --steps 20 --warmup 5 --windows 1 --fresh-batches 0 --no-cpu-baseline > "$O/kstats_b4096.txt"
python3 tools/step_timeline.py "$O/prof_b4096" > "$O/step_timeline_graph.txt"
find "$O/prof_b4096" -name "*kernel_trace.csv" -delete     # (gpurun brings back at most 64 MiB: the summaries stay, the raw traces go)
# 3. the fresh-batch graph's timeline
tools/prof.sh "$tag/prof_fresh" bench.py --steps 10 --warmup 3 --windows 1 --fresh-batches 16 --no-cpu-baseline > "$O/kstats_fresh.txt"
python3 tools/step_timeline.py "$O/prof_fresh" > "$O/step_timeline_fresh.txt"
find "$O/prof_fresh" -name "*kernel_trace.csv" -delete
# 4. configs[2]: AID 435008 shape, batch 256
MKGNN_NO_SMALL_BATCH= MKGNN_NO_EXACT_LEG= python3 bench.py --assay 435008 --batch-size 256 --steps 200 --warmup 20 > "$O/bench_435008_b256.json" 2> "$O/bench_435008_b256.err"
tools/prof.sh "$tag/prof_b256" bench.py --assay 435008 --batch-size 256 --steps 50 --warmup 5 --windows 1 --fresh-batches 0 --no-cpu-baseline > "$O/kstats_435008_b256.txt"
python3 tools/step_timeline.py "$O/prof_b256" > "$O/step_timeline_435008_b256.txt"
find "$O/prof_b256" -name "*kernel_trace.csv" -delete
# 5. bf16 similarity variant (configs[4] shape on one GPU)
python3 bench.py --variant bf16 --steps 50 --warmup 10 --no-cpu-baseline --fresh-batches 0 > "$O/bench_b4096_bf16.json" 2> "$O/bench_b4096_bf16.err"
python3 bench.py --assay all9 --variant bf16 --steps 50 --warmup 10 --no-cpu-baseline --fresh-batches 0 > "$O/bench_all9_bf16.json" 2> "$O/bench_all9_bf16.err"
python3 tools/shard_loader_probe.py --molecules 131072 --workers 2 > "$O/shard_loader.txt" 2>&1 || echo "shard_loader_probe failed" >> "$O/shard_loader.txt"
fi
if [ "$part" != rest ]; then
# 6. HBM traffic of the forward kernels (separate PMC passes) -> the JSONs bench.py reads: the N-hop layer at batch 4096
#    (roofline.traffic), the 1-hop layer at batch 4096 (roofline.kernels), the N-hop layer at configs[2]'s batch of 256
pmc_fwd() {      # <name> <kernel substring> <json> <workload text> <fwd_probe arguments ...>
    local name="$1" kern="$2" json="$3" what="$4"; shift 4
    tools/pmc.sh "$tag/$name" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" -- tools/fwd_probe.py --reps 6 "$@" > "$O/$name.txt"
    local alg
    alg=$(python3 tools/fwd_probe.py --reps 1 "$@" 2>/dev/null | grep -o "algorithmic_bytes=[0-9]*" | cut -d= -f2)
    python3 tools/collect_pmc.py "$kern" "$O/$json" "$commit" "$alg" "$O/$name" "$what" > /dev/null
    rm -rf "$O/$name"/pass*/
}
pmc_fwd pmc_fwd "kc_forward_stream<7" forward_pmc.json "tools/fwd_probe.py (batch 4096 molecules, ~102.5 k atoms, N-hop layer F=110, training configuration)"
pmc_fwd pmc_fwd_1hop "kc_forward_stream<2" forward_pmc_1hop.json "tools/fwd_probe.py --width 28 (batch 4096 molecules, ~102.5 k atoms, 1-hop layer F=28, training configuration)" --width 28
pmc_fwd pmc_fwd_b256 "kc_forward_stream<7" forward_pmc_b256.json "tools/fwd_probe.py --batch-size 256 (256 molecules, ~6.4 k atoms, N-hop layer F=110, training configuration: BASELINE configs[2]'s batch)" --batch-size 256
fi
if [ "$part" != pmc ]; then
# 7. pipe utilisation counters of every kernel of a step
tools/pmc.sh "$tag/pmc_step" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
    "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
    "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
    "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max" "FETCH_SIZE" "WRITE_SIZE" \
    -- bench.py --steps 3 --warmup 1 --windows 1 --fresh-batches 0 --no-cpu-baseline --roofline-reps 2 > "$O/pmc_step.txt"
rm -rf "$O"/pmc_step/pass*/
fi
du -sh "$O"
echo done
