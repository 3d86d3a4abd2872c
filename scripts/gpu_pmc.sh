#!/bin/bash
# HBM traffic of the hot kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2, they do not fit one pass), kernel-trace only.
tag=${1:-r01}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o pmc -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/pmc_${tag}_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 gpurun_out/pmc_${tag}_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f:
        print(c, "no counter file"); continue
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") == c:
            acc[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if "anonymous" in k:
            print(f"{c:10s} {k:60s} launches {len(v):3d} mean {sum(v)/len(v):14.1f}")
PY
