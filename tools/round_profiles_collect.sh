#!/bin/bash
# Copy the artefacts tools/round_profiles.sh left under gpurun_out/<tag>/ into profiles/ as <prefix>_*:
# tools/round_profiles_collect.sh <tag> <prefix, e.g. r02>
set -e
O="gpurun_out/$1"; P="profiles/$2"
tail -1 "$O/bench_b4096.json" > "${P}_bench_b4096.json"
tail -1 "$O/bench_435008_b256.json" > "${P}_bench_435008_b256.json"
tail -1 "$O/bench_b4096_bf16.json" > "${P}_bench_b4096_bf16.json"
tail -1 "$O/bench_all9_bf16.json" > "${P}_bench_all9_bf16.json"
grep -v "amdgpu.ids" "$O/shard_loader.txt" > "${P}_shard_loader.txt"
cp "$(ls -t $O/prof_b4096/*/*kernel_stats.csv | head -1)" "${P}_bench_b4096_kernel_stats.csv"
cp "$(ls -t $O/prof_fresh/*/*kernel_stats.csv | head -1)" "${P}_bench_b4096_fresh_kernel_stats.csv"
cp "$(ls -t $O/prof_b256/*/*kernel_stats.csv | head -1)" "${P}_bench_435008_b256_kernel_stats.csv"
cp "$O/step_timeline_graph.txt" "${P}_step_timeline_graph.txt"
cp "$O/step_timeline_fresh.txt" "${P}_step_timeline_fresh_graph.txt"
cp "$O/step_timeline_435008_b256.txt" "${P}_step_timeline_435008_b256.txt"
cp "$O/forward_pmc.json" "${P}_forward_pmc.json"
cp "$O/forward_pmc_1hop.json" "${P}_forward_pmc_1hop.json"
cp "$O/forward_pmc_b256.json" "${P}_forward_pmc_b256.json"
grep -v "^W2026\|^E2026" "$O/pmc_step.txt" > "${P}_pmc_step_kernels.txt"
ls -la profiles | grep "$2"
