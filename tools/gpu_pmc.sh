#!/bin/bash
# PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes) at EVERY call shape bench.py may be asked for:
# PMC_SHAPES (default "20 32") requests per mldhip_sample_many call -- 20 = the driver's `bench.py --steps 20`, 32 = the chip-filling call.
# Per shape: FETCH_SIZE and WRITE_SIZE per kernel, then the SQ counters that say where the time goes.  Writes ONE summary,
# gpurun_out/<TAG>_pmc_traffic.json = {"shapes": {"20": {...}, "32": {...}}}, each entry stamped with the hash of the engine sources AND of the
# loop kernel's machine code it ran on (bench.py refuses an entry that matches neither; shape 1 = ONE bs-64 request: the cluster loop).  Copy it to profiles/r05_pmc_traffic.json.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r05}
SHAPES=${PMC_SHAPES:-"1 20 32"}
PASSES=("FETCH_SIZE" "WRITE_SIZE")
[ "${PMC_TRAFFIC_ONLY:-0}" = 1 ] || PASSES+=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
for COAL in $SHAPES; do
  for C in "${PASSES[@]}"; do
    T=$(echo $C | tr ' ' '_' | cut -c1-48)
    cd /tmp && timeout ${PMC_TIMEOUT:-180} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c${COAL}_$T -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --profile-child --precision ${PMC_PRECISION:-f16x3} --coalesce $COAL --steps 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_c${COAL}_$T.log 2>&1
    cd $GRAFT_REPO_ROOT
  done
done
python - "$TAG" $SHAPES <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
tag, shapes = sys.argv[1], sys.argv[2:]
NAMES = {"den_loop": "den_loop_kernel", "den_cluster": "den_cluster_kernel", "dec_ffn": "ffn_strip_x3_kernel", "dec_qkv": "strip_gemm_x3_kernel<6, 1, false, true",
         "dec_skip": "strip_gemm_x3_kernel<4, 2, false, false", "dec_attn": "attn_flash_x3_kernel", "dec_final": "final_strip_x3_kernel"}
out = {"note": ("rocprofv3 --pmc, separate passes with --kernel-trace only, over bench.py --profile-child --coalesce N --steps 1 (one call of N bs-64 requests).  "
                "FETCH/WRITE_SIZE are KB; fetch bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read correction of MI355X_MICROARCH.md); weights and the working set of the loop "
                "are Infinity-Cache resident, so the loop's number is L2<->fabric traffic, not DRAM traffic.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, "
                "SQ_VALU_MFMA_BUSY_CYCLES cycles (16 per v_mfma_f32_16x16x32_f16); GRBM_GUI_ACTIVE is summed over the 8 XCDs."), "shapes": {}}
for coal in shapes:
    agg = collections.defaultdict(dict)          # kernel -> counter -> (dispatches, average per dispatch)
    for f in glob.glob(f"gpurun_out/pmc_{tag}_c{coal}_*/**/*counter_collection*.csv", recursive=True):
        a = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            e = a[(row["Kernel_Name"], row["Counter_Name"])]; e[0] += 1; e[1] += float(row["Counter_Value"])
        for (k, c), (n, v) in a.items():
            agg[k][c] = (n, v / n)
    traffic, sq = {}, {}
    for short, pat in NAMES.items():
        hits = [k for k in agg if pat in k]
        if not hits:
            continue
        k = max(hits, key=lambda kk: max(v[1] for v in agg[kk].values()))       # (the big launches, not a tiny one-off of the same template)
        ent = {c: v[1] for c, v in agg[k].items()}
        ent["kernel"], ent["dispatches"] = k[:100], max(v[0] for v in agg[k].values())
        if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
            ent["traffic_bytes_per_launch"] = int(2 * ent["FETCH_SIZE"] * 1024 + ent["WRITE_SIZE"] * 1024)
        traffic[short] = {c: ent[c] for c in ("kernel", "dispatches", "FETCH_SIZE", "WRITE_SIZE", "traffic_bytes_per_launch") if c in ent}
        s = {c: ent[c] for c in ent if c.startswith("SQ_") or c.startswith("GRBM")}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s:
            s["mfma_busy_frac"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (s["GRBM_GUI_ACTIVE"] / 8), 4)      # cycles over the 1 024 SIMDs / cycles per XCD
        if "SQ_INSTS_VALU" in s and "SQ_INSTS_MFMA" in s and s["SQ_INSTS_MFMA"]:
            s["valu_per_mfma"] = round((s["SQ_INSTS_VALU"] - s["SQ_INSTS_MFMA"]) / s["SQ_INSTS_MFMA"], 2)
        if "SQ_LDS_BANK_CONFLICT" in s and s.get("SQ_LDS_IDX_ACTIVE"):
            s["lds_conflict_frac"] = round(s["SQ_LDS_BANK_CONFLICT"] / s["SQ_LDS_IDX_ACTIVE"], 4)
        if len(s) > 1:
            sq[short] = dict(kernel=k[:100], **s)
    loop = next((k for k in agg if "den_loop_kernel" in k), None) or next((k for k in agg if "den_cluster_kernel" in k), None)
    out["shapes"][str(coal)] = {"source_hash": bench.source_hash(), "loop_kernel_code_hash": bench.kernel_code_hash(bench.mangled_part(loop)) if loop else None,
                                "loop_kernel": loop[:80] if loop else None,
                                "requests_per_call": int(coal), "kernels": traffic, "sq": sq}
json.dump(out, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
for coal, e in out["shapes"].items():
    print(coal, json.dumps({k: v.get("traffic_bytes_per_launch") for k, v in e["kernels"].items()}), json.dumps({k: {c: v[c] for c in ("mfma_busy_frac", "valu_per_mfma", "lds_conflict_frac") if c in v} for k, v in e["sq"].items()}))
for f in glob.glob("gpurun_out/pmc_*/**/*.csv", recursive=True):
    if os.path.getsize(f) > 4 << 20: os.remove(f)
PY
