#!/bin/bash
# HBM traffic of the hot kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2, they do not fit one pass), kernel-trace only.
tag=${1:-r02}
wl=${2:-pure}        # pure | hybrid: which bench workload (output: gpurun_out/<tag>_pmc_hbm[_hybrid].json)
sfx=""; [ "$wl" != "pure" ] && sfx="_$wl"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}${sfx}_$c -o pmc -- python3 bench.py --no-cpu-baseline --no-extras --workload $wl --steps 5 --warmup 2 --clock-settle-ms 0 > gpurun_out/pmc_${tag}${sfx}_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 gpurun_out/pmc_${tag}${sfx}_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json, re
def _name(raw):
    # (rocprofv3 leaves names with a bf16 / fp16 template argument mangled, and binutils' c++filt does not know DF16b)
    m = re.search(r"_GLOBAL__N_1\\d+(\\w+_kernel)I((?:Li\\d+E)*)(DF16b|DF16_)?E", raw)
    if raw.startswith("_Z") and m:
        args = re.findall(r"Li(\\d+)E", m.group(2))
        if m.group(3):
            args.append("__bf16" if m.group(3) == "DF16b" else "_Float16")
        return "%s<%s>" % (m.group(1), ", ".join(args))
    return re.sub(r"^(void )?\\(anonymous namespace\\)::", "", raw).split("(")[0]
out = {"source": "rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-extras --workload ${wl} --steps 5 --warmup 2 --clock-settle-ms 0 "
                 "(two passes, scripts/gpu_pmc.sh), MI355X, tag ${tag}",
       "units": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; gfx950 FETCH_SIZE counts 128-B read requests as 64 B "
                "(MI355X_MICROARCH.md HBM section): hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_${tag}${sfx}_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f:
        print(c, "no counter file"); continue
    acc, grid = collections.defaultdict(list), collections.defaultdict(int)
    rows = [r for r in csv.DictReader(open(f[0])) if r.get("Counter_Name") == c]
    for row in rows:                       # a kernel's full-size launches only (the bench also times a lone workgroup of the sweep)
        grid[_name(row["Kernel_Name"])] = max(grid[_name(row["Kernel_Name"])], int(row["Grid_Size"]))
    for row in rows:
        name = _name(row["Kernel_Name"])
        if int(row["Grid_Size"]) == grid[name]:
            acc[name].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if "at::" in k or "elementwise" in k:
            continue
        d = out["kernels"].setdefault(k, {})
        d[c + "_KiB_mean"] = sum(v) / len(v)
        d["launches"] = len(v)
        print(f"{c:10s} {k:60s} launches {len(v):3d} mean {sum(v)/len(v):14.1f} KiB")
for k, d in out["kernels"].items():
    d["hbm_bytes_corrected"] = (2 * d.get("FETCH_SIZE_KiB_mean", 0.0) + d.get("WRITE_SIZE_KiB_mean", 0.0)) * 1024
json.dump(out, open("gpurun_out/${tag}_pmc_hbm${sfx}.json", "w"), indent=1)
PY
