#!/bin/bash
# rocprofv3 kernel stats of the device training step (scripts/time_train.py).  usage: scripts/prof_train.sh <tag>
tag=${1:-train}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$tag -o train -- python3 scripts/time_train.py > gpurun_out/prof_train_$tag.log 2>&1 || { echo "rocprof failed"; tail -20 gpurun_out/prof_train_$tag.log; exit 1; }
tail -5 gpurun_out/prof_train_$tag.log
f=$(find gpurun_out/prof_train_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" | cut -c1-200
