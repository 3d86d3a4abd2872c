#!/bin/bash
# pure vs hybrid iteration over batch sizes (one box): where, if anywhere, does the transformer-predicted iteration win?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for wl in pure hybrid; do
  for b in 256 1024 4096 16384 32768; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --workload $wl --batch $b --steps 20 > gpurun_out/bb.json 2> gpurun_out/bb.err || { echo "$wl $b failed"; tail -3 gpurun_out/bb.err; continue; }
    python3 -c "
import json; d=json.load(open('gpurun_out/bb.json'))
print('$wl', $b, 'ms/iter %.4f' % d['ms_per_step'], 'steps/s %.3e' % d['value'], {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, 'roof %.3f' % d['roofline']['frac'])"
  done
done
