#!/bin/bash
# SQ-level counters of the hot kernels (where do the sweep's cycles go?): one rocprofv3 --pmc pass per counter group,
# kernel-trace only (never combined with sys/hip/hsa traces).  Usage: scripts/gpu_pmc_sq.sh <tag>
tag=${1:-r01}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/sq_${tag}_$i -o pmc -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/sq_${tag}_$i.log 2>&1 || { echo "pmc group $i ($grp) failed"; tail -3 gpurun_out/sq_${tag}_$i.log; continue; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_${tag}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        for short in ("sweep_tile16", "linearize_euler", "linesearch_quad", "simulate_quad"):
            if short in k:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} launches {len(v):3d} mean {sum(v)/len(v):16.1f}")
PY
