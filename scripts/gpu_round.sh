#!/bin/bash
# One GPU-box visit: parity tests, headline bench, rocprofv3 kernel stats of the same bench command.
# usage: scripts/gpu_round.sh <tag> [skip-tests]
tag=${1:-r02}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{ echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "v1 quota $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1)";
  cat /proc/self/cgroup; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; } > gpurun_out/host_$tag.txt 2>&1
if [ "$2" != "skip-tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$tag.log 2>&1
  rc=$?
  tail -15 gpurun_out/pytest_$tag.log
  if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc: stopping"; exit 1; fi
fi
timeout -k 10 400 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { echo "bench failed"; tail -20 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof failed"; tail -20 gpurun_out/prof_$tag.log; exit 1; }
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-220
# hybrid workload (BASELINE configs[4]): kernel stats of the same command the bench's extras run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hybrid_$tag -o bench -- python3 bench.py --no-cpu-baseline --no-extras --workload hybrid --steps 100 --warmup 20 > gpurun_out/prof_hybrid_$tag.log 2>&1 || { echo "rocprof hybrid failed"; tail -20 gpurun_out/prof_hybrid_$tag.log; exit 1; }
f=$(find gpurun_out/prof_hybrid_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-220
