#!/bin/bash
# SQ-level counters of the hot kernels: one rocprofv3 --pmc pass per counter group, kernel-trace only (never combined with
# sys/hip/hsa traces).  Usage: scripts/gpu_pmc_sq.sh <tag> [pure|hybrid]   -> gpurun_out/<tag>_pmc_sq_<workload>.json
tag=${1:-r02}
wl=${2:-pure}
cmd=${3:-"bench.py --no-cpu-baseline --no-extras --workload $wl --steps 3 --warmup 1 --clock-settle-ms 0"}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/sq_${tag}_${wl}_$i -o pmc -- python3 $cmd > gpurun_out/sq_${tag}_${wl}_$i.log 2>&1 || { echo "pmc group $i ($grp) failed"; tail -3 gpurun_out/sq_${tag}_${wl}_$i.log; continue; }
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
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_${tag}_${wl}_*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    grid = collections.defaultdict(int)    # a kernel's full-size launches only (the bench also times a lone workgroup of the sweep)
    for row in rows:
        grid[_name(row["Kernel_Name"])] = max(grid[_name(row["Kernel_Name"])], int(row["Grid_Size"]))
    for row in rows:
        k = _name(row["Kernel_Name"])
        if "at::" in k or "elementwise" in k or "rocclr" in k or "Cat" in k or int(row["Grid_Size"]) != grid[k]:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"source": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --no-extras --workload ${wl} "
                 "--steps 3 --warmup 1 --clock-settle-ms 0 (one pass per counter group, scripts/gpu_pmc_sq.sh), MI355X, tag ${tag}",
       "units": "means per launch, summed over the chip; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, "
                "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md); derived: valu_per_wave, mfma_per_wave, "
                "mfma_busy_frac = VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES summed over 4 SIMDs), wait_frac = WAIT_ANY / WAVE_CYCLES",
       "kernels": {}}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = m.get("SQ_WAVES", 0) or 1
    der = {"valu_per_wave": m.get("SQ_INSTS_VALU", 0) / w, "mfma_per_wave": m.get("SQ_INSTS_MFMA", 0) / w,
           "lds_per_wave": m.get("SQ_INSTS_LDS", 0) / w, "salu_per_wave": m.get("SQ_INSTS_SALU", 0) / w}
    if m.get("SQ_WAVE_CYCLES"):
        der["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
        der["wait_inst_any_frac"] = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
        der["active_inst_any_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"]
        der["active_valu_frac"] = m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"]
    if m.get("SQ_BUSY_CYCLES"):
        der["mfma_busy_over_sq_busy"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / m["SQ_BUSY_CYCLES"]
    out["kernels"][k] = {"counters": m, "derived": der, "launches": max(len(v) for v in d.values())}
    print(k)
    for c, v in sorted(m.items()):
        print(f"   {c:32s} {v:18.1f}")
    for c, v in der.items():
        print(f"   -> {c:29s} {v:18.4f}")
json.dump(out, open("gpurun_out/${tag}_pmc_sq_${wl}.json", "w"), indent=1)
PY
