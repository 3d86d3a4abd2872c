#!/bin/bash
# Round-4 GPU-box visit: parity tests, the full bench line, rocprofv3 kernel stats of the bench commands (pure + hybrid), PMC passes
# (HBM bytes pure + hybrid; SQ counters pure + hybrid) and the single-trajectory latency profile.  usage: scripts/gpu_round4.sh <tag>
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{ echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>&1)"; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; } > gpurun_out/host_$tag.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_$tag.log
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc: stopping"; exit 1; fi
timeout -k 10 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { echo "bench failed"; tail -20 gpurun_out/bench_$tag.err; exit 1; }
echo "bench done"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof failed"; tail -20 gpurun_out/prof_$tag.log; exit 1; }
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hybrid_$tag -o bench -- python3 bench.py --no-cpu-baseline --no-extras --workload hybrid --steps 100 --warmup 20 > gpurun_out/prof_hybrid_$tag.log 2>&1 || { echo "rocprof hybrid failed"; tail -20 gpurun_out/prof_hybrid_$tag.log; exit 1; }
f=$(find gpurun_out/prof_hybrid_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
# single-trajectory drop-in (the reference's use case): kernel stats of repeated optimize() calls
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b1_$tag -o b1 -- python3 scripts/prof_dropin_latency.py quadrotor log > gpurun_out/prof_b1_$tag.log 2>&1 || { echo "rocprof b1 failed"; tail -5 gpurun_out/prof_b1_$tag.log; }
f=$(find gpurun_out/prof_b1_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -5 "$f" | cut -c1-200
scripts/gpu_pmc.sh $tag pure || exit 1
scripts/gpu_pmc.sh $tag hybrid || exit 1
scripts/gpu_pmc_sq.sh $tag pure > gpurun_out/pmc_sq_pure_$tag.log 2>&1 || { echo "pmc sq pure failed"; tail -5 gpurun_out/pmc_sq_pure_$tag.log; }
scripts/gpu_pmc_sq.sh $tag hybrid > gpurun_out/pmc_sq_hybrid_$tag.log 2>&1 || { echo "pmc sq hybrid failed"; tail -5 gpurun_out/pmc_sq_hybrid_$tag.log; }
ls gpurun_out | grep "${tag}_pmc"
