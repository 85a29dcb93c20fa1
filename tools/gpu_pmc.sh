#!/bin/bash
# PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes): FETCH_SIZE and WRITE_SIZE per kernel of
# the headline workload (one step in flight), then SQ_VALU_MFMA_BUSY_CYCLES.  Writes gpurun_out/<TAG>_pmc_traffic.json stamped
# with the hash of the engine sources it ran on (bench.py refuses a summary whose hash differs from the build it benches).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
# (the TCC passes hung -- 900 s and 150 s timeouts -- when the child made FOUR coalesced calls; with ONE call (capture + first replay,
#  450 launches of every loop kernel) each pass takes seconds.  Keep one call, the runtime's default of 4 hardware queues, a short timeout.)
export GPU_MAX_HW_QUEUES=${PMC_HW_QUEUES:-4}
for C in ${PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES}; do
  cd /tmp && timeout ${PMC_TIMEOUT:-180} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C -o pmc -- \
    python $GRAFT_REPO_ROOT/bench.py --profile-child --precision ${PMC_PRECISION:-f16x3} --coalesce ${PMC_COALESCE:-5} --steps ${PMC_CALLS:-1} > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C.log 2>&1
  cd $GRAFT_REPO_ROOT
done
python - "$TAG" <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
tag = sys.argv[1]
agg = {}
for C in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    files = glob.glob(f"gpurun_out/pmc_{tag}_{C}/**/*counter_collection*.csv", recursive=True)
    if not files:
        print("no counter file for", C); continue
    a = collections.defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != C: continue
            a[row["Kernel_Name"]][0] += 1; a[row["Kernel_Name"]][1] += float(row["Counter_Value"])
    agg[C] = {k: (n, v / n) for k, (n, v) in a.items()}
kern = {}
coalesce = int(os.environ.get("PMC_COALESCE", "5"))
for name, (prefix, _) in bench.kernel_table(bench.BATCH * coalesce, os.environ.get("PMC_PRECISION", "f16x3")).items():
    ent = {}
    for C, table in agg.items():
        hits = [(k, v) for k, v in table.items() if k.startswith(prefix)]
        if name.startswith("dec_") and name not in ("dec_attn", "dec_ffn"):
            kcs = ", 32, false>" if name == "dec_ffn2_ln" else ", 8, false>"
            hits = [(k, v) for k, v in hits if kcs in k]
        if hits:
            k, (n, avg) = max(hits, key=lambda kv: kv[1][0])
            ent[C] = avg; ent["kernel"] = k[:90]; ent["dispatches"] = n
    if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
        # counters are KB; gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section)
        ent["traffic_bytes_per_launch"] = int(2 * ent["FETCH_SIZE"] * 1024 + ent["WRITE_SIZE"] * 1024)
    kern[name] = ent
out = {"source_hash": bench.source_hash(), "requests_per_call": coalesce,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate passes with --kernel-trace only, over bench.py --profile-child "
               "--coalesce N (the headline call shape, one call in flight).  FETCH/WRITE_SIZE are KB; fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read correction of "
               "MI355X_MICROARCH.md); the working set is Infinity-Cache resident, so this is L2<->fabric traffic, not DRAM traffic.",
       "kernels": kern}
json.dump(out, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
for f in glob.glob("gpurun_out/pmc_*/**/*.csv", recursive=True):
    if os.path.getsize(f) > 4 << 20: os.remove(f)
PY
