#!/bin/bash
# A/B on ONE box: the headline bench (kernel times by HIP events) with the shipped library and with every library given
# (absolute or repo-relative paths, e.g. build_ab/lib_old.so), two rounds, interleaved.
# usage: scripts/ab_lib.sh [--workload pure|hybrid|cartpole] lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
wl=pure
if [ "$1" == "--workload" ]; then wl=$2; shift 2; fi
show='import sys,json; d=json.loads(sys.stdin.readline()); print("%.4f ms/step" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernel_ms"].items()})'
for rep in 1 2; do
  echo -n "[shipped] "; timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --workload $wl --steps 200 --warmup 20 2>/dev/null | python3 -c "$show" || exit 1
  for l in "$@"; do
    echo -n "[$l] "; QUATTRO_HIP_LIB=$(realpath $l) timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --workload $wl --steps 200 --warmup 20 2>/dev/null | python3 -c "$show" || exit 1
  done
done
