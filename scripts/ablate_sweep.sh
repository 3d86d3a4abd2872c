#!/bin/bash
# Ablation of the fused sweep's step (csrc/sweep_tile16_body.h, -DQT_ABLATE=n): builds one library per left-out segment into
# build_ab/ (in the build container: hipcc cross-compiles), then, on the GPU box, scripts/ab_lib.sh build_ab/lib_abl*.so times the
# headline bench with each — `sweep` (B = 4096, full load) and the roofline block's lone_wave_sweep_ms.
# usage (build container): scripts/ablate_sweep.sh build       usage (GPU box): scripts/ablate_sweep.sh run
cd "$(dirname "$0")/.." || exit 1
C=quattro-transformer-ilqr_amd/csrc
if [ "$1" == "build" ]; then
  mkdir -p build_ab
  for n in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -DQT_ABLATE=$n \
      -c $C/sweep_tile16.hip -o build_ab/sweep_tile16_abl$n.o || exit 1
    objs=$(ls $C/*.o | grep -v sweep_tile16.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_ab/sweep_tile16_abl$n.o -o build_ab/lib_abl$n.so || exit 1
  done
  ls -la build_ab/lib_abl*.so
else
  show='import sys,json; d=json.loads(sys.stdin.readline()); print("sweep %.1f us, lone wave %.1f us" % (1e3*d["kernel_ms"]["sweep"], 1e3*d["roofline"]["lone_wave_sweep_ms"]))'
  for rep in 1 2; do
    echo -n "[shipped] "; python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 2>/dev/null | python3 -c "$show"
    for n in 1 2 3 4 5; do
      echo -n "[without segment $n] "; QUATTRO_HIP_LIB=$(realpath build_ab/lib_abl$n.so) python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 2>/dev/null | python3 -c "$show"
    done
  done
fi
