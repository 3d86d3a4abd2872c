#!/bin/bash
# Ablation of the quadrotor rollout step (csrc/rollout_quad_body.h, -DQT_ABLATE_LS=n): one library per left-out segment into
# build_ab/ (build container), then on the GPU box the headline bench's `simulate` and `linesearch` kernel times with each.
# usage (build container): scripts/ablate_linesearch.sh build      usage (GPU box): scripts/ablate_linesearch.sh run
cd "$(dirname "$0")/.." || exit 1
C=quattro-transformer-ilqr_amd/csrc
if [ "$1" == "build" ]; then
  mkdir -p build_ab
  for n in 1 2 3 4 5 6; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DQT_ABLATE_LS=$n \
      -c $C/rollout_quad.hip -o build_ab/rollout_quad_abl$n.o || exit 1
    objs=$(ls $C/*.o | grep -v rollout_quad.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_ab/rollout_quad_abl$n.o -o build_ab/lib_lsabl$n.so || exit 1
  done
  ls build_ab/lib_lsabl*.so
else
  show='import sys,json; d=json.loads(sys.stdin.readline()); print("simulate %.1f us, line search %.1f us" % (1e3*d["kernel_ms"]["simulate"], 1e3*d["kernel_ms"]["linesearch"]))'
  for rep in 1 2; do
    echo -n "[shipped] "; python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 2>/dev/null | python3 -c "$show"
    for n in 1 2 3 4 5 6; do
      echo -n "[without segment $n] "; QUATTRO_HIP_LIB=$(realpath build_ab/lib_lsabl$n.so) python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 2>/dev/null | python3 -c "$show"
    done
  done
fi
