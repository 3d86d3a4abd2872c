#!/bin/bash
# A/B on ONE box: the shipped library vs variants of csrc/tf_stream.hip built with extra -D flags (one per argument).
# usage: scripts/ab_tf.sh "-DQT_TF_VAR_X" "-DQT_TF_VAR_Y=2" ...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab
C=quattro-transformer-ilqr_amd/csrc
i=0
libs=("")
for fl in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize $fl -c $C/tf_stream.hip -o gpurun_out/ab/tf_$i.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/capi.o $C/sweep_generic.o $C/sweep_tile16.o $C/sweep_lane.o $C/linearize.o $C/rollout.o $C/rollout_quad.o $C/tf_train.o gpurun_out/ab/tf_$i.o -o gpurun_out/ab/lib_$i.so || exit 1
  libs+=("$PWD/gpurun_out/ab/lib_$i.so")
done
for rep in 1 2; do
  j=0
  for l in "${libs[@]}"; do
    if [ -z "$l" ]; then timeout -k 10 120 python scripts/time_tf.py; else echo -n "[${@:$j:1}] "; QUATTRO_HIP_LIB=$l timeout -k 10 120 python scripts/time_tf.py; fi
    j=$((j+1))
  done
done
