#!/bin/bash
# copy the summaries of a GPU-box visit (scripts/gpu_round.sh <tag>) from gpurun_out/ (scratch) into profiles/ (tracked)
tag=$1
cd "$(dirname "$0")/.." || exit 1
cp gpurun_out/bench_$tag.json profiles/${tag}_bench.json 2>/dev/null
cp gpurun_out/prof_$tag/bench_kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_hybrid_$tag/bench_kernel_stats.csv profiles/${tag}_bench_hybrid_kernel_stats.csv 2>/dev/null
cp gpurun_out/host_$tag.txt profiles/${tag}_host.txt 2>/dev/null
tail -3 gpurun_out/pytest_$tag.log > profiles/${tag}_pytest_tail.txt 2>/dev/null
ls profiles | grep "^$tag"
cp gpurun_out/prof_b1_$tag/b1_kernel_stats.csv profiles/${tag}_dropin_b1_kernel_stats.csv 2>/dev/null
for f in pmc_hbm pmc_hbm_hybrid pmc_sq_pure pmc_sq_hybrid; do cp gpurun_out/${tag}_$f.json profiles/${tag}_$f.json 2>/dev/null; done
ls profiles | grep "^$tag"
