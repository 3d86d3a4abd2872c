#!/bin/bash
# Round-3 GPU-box visit: scripts/gpu_round.sh (tests, bench, rocprof of the bench commands) + kernel stats of the device-resident
# loops + PMC passes (HBM bytes of the headline kernels; SQ counters of the persistent kernels).  usage: scripts/gpu_round3.sh <tag>
tag=${1:-r03}
cd "$GRAFT_REPO_ROOT" || exit 1
scripts/gpu_round.sh $tag || exit 1
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_loops_$tag -o loops -- python3 scripts/prof_device_loop.py > gpurun_out/prof_loops_$tag.log 2>&1 || { echo "rocprof device loops failed"; tail -20 gpurun_out/prof_loops_$tag.log; exit 1; }
f=$(find gpurun_out/prof_loops_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-200
scripts/gpu_pmc.sh $tag pure || exit 1
scripts/gpu_pmc_sq_cmd.sh $tag loops "scripts/prof_device_loop.py" > gpurun_out/pmc_sq_loops_$tag.log 2>&1 || { echo "pmc sq loops failed"; tail -5 gpurun_out/pmc_sq_loops_$tag.log; }
tail -40 gpurun_out/pmc_sq_loops_$tag.log | head -60
# user-compiled model: kernel stats of the generic path
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_user_$tag -o user -- python3 scripts/time_user_model.py > gpurun_out/prof_user_$tag.log 2>&1 || { echo "rocprof user model failed"; tail -20 gpurun_out/prof_user_$tag.log; }
grep planar gpurun_out/prof_user_$tag.log
f=$(find gpurun_out/prof_user_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-200
