#!/bin/bash
# One more PMC pass (kernel-trace only): MFMA pipe busy cycles and the kernel's active GPU cycles -> MfmaUtil per kernel
# (gfx94x formula, MI355X_MICROARCH.md: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4 SIMDs)).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-pmc_mfma}
cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG} -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-a2m --no-novae --no-clip --in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json
files = glob.glob("gpurun_out/${TAG}/*counter_collection*.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(files[0]) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"][:90]
        a = agg[k][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
out = {}
for k, c in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c: continue
    n = c["GRBM_GUI_ACTIVE"][0]
    busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n, c["GRBM_GUI_ACTIVE"][1] / n
    out[k] = {"dispatches": n, "mfma_busy_cycles_per_dispatch": busy, "gui_active_cycles_per_dispatch": act,
              "mfma_util_256cu_x4": busy / (act * 256 * 4) if act else None}
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_dispatch"] * kv[1]["dispatches"])[:14]:
    print(k, v)
json.dump(out, open("gpurun_out/${TAG}_summary.json", "w"), indent=1)
PY
for f in gpurun_out/${TAG}/*.csv; do [ $(stat -c %s "$f") -gt 8388608 ] && rm -f "$f"; done
tail -2 gpurun_out/${TAG}.log
