#!/bin/bash
# PMC passes (separate runs, kernel-trace only): FETCH_SIZE and WRITE_SIZE per kernel.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-pmc}
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-a2m --no-novae --no-clip --in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C.log 2>&1
  cd $GRAFT_REPO_ROOT
  ls gpurun_out/pmc_${TAG}_$C | head
done
python - <<'PY'
import csv, glob, collections, json, os
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_*_{C}/*counter_collection*.csv")
    if not files:
        print("no counter file for", C, glob.glob(f"gpurun_out/pmc_*_{C}/*")); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != C: continue
            k = row["Kernel_Name"][:70]
            agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    out[C] = {k: {"dispatches": n, "avg_per_dispatch": v / n} for k, (n, v) in agg.items()}
    for k, d in sorted(out[C].items(), key=lambda kv: -kv[1]["avg_per_dispatch"] * kv[1]["dispatches"])[:12]:
        print(C, k, d)
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
# keep the merged payload small
for f in glob.glob("gpurun_out/pmc_*/*.csv"):
    if os.path.getsize(f) > 8 << 20: os.remove(f)
PY
