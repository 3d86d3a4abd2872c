#!/bin/bash
# transformer kernel iteration: parity tests of the predictor, then the hybrid bench line (no CPU leg)
tag=${1:-tf}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 240 python -m pytest tests/test_transformer_gpu.py -m gpu -q -x -s > gpurun_out/pytest_tf_$tag.log 2>&1
rc=$?
tail -25 gpurun_out/pytest_tf_$tag.log
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc: stopping"; exit 1; fi
timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --workload hybrid --steps 20 > gpurun_out/bench_hybrid_$tag.json 2> gpurun_out/bench_hybrid_$tag.err || { echo "hybrid bench failed"; tail -20 gpurun_out/bench_hybrid_$tag.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_hybrid_$tag.json')); print(d['kernel_ms'], d['roofline']['frac'], d['ms_per_step'])"
