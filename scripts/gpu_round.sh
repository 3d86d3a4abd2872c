#!/bin/bash
# One GPU-box visit: parity tests, headline bench, rocprofv3 kernel stats of the same bench command.
# usage: scripts/gpu_round.sh <tag>
tag=${1:-r01}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_$tag.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
timeout -k 10 300 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { echo "bench failed"; tail -20 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof failed"; tail -20 gpurun_out/prof_$tag.log; exit 1; }
find gpurun_out/prof_$tag -name "*stats*" | head
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f"
# hybrid workload (BASELINE configs[4]): bench line + kernel stats
timeout -k 10 300 python bench.py --no-cpu-baseline --workload hybrid > gpurun_out/bench_hybrid_$tag.json 2> gpurun_out/bench_hybrid_$tag.err || { echo "hybrid bench failed"; tail -20 gpurun_out/bench_hybrid_$tag.err; exit 1; }
cat gpurun_out/bench_hybrid_$tag.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hybrid_$tag -o bench -- python3 bench.py --no-cpu-baseline --workload hybrid --steps 20 > gpurun_out/prof_hybrid_$tag.log 2>&1 || { echo "rocprof hybrid failed"; tail -20 gpurun_out/prof_hybrid_$tag.log; exit 1; }
f=$(find gpurun_out/prof_hybrid_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-200
