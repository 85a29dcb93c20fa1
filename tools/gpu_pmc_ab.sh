#!/bin/bash
# SQ counters of the decoder's kernels at the headline call shape with the late round-3 options ON (defaults) and OFF
# ("attn_tr" = 0, "nt_hints" = 0, "dec_l0_once" = 0): LDS bank conflicts / LDS instructions / MFMA-busy of the key-blocked attention
# with V through ds_read_b64_tr_b16 against the transposed-plane form.  Separate rocprofv3 --pmc passes, --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for CFG in "on:" "off:attn_tr=0,nt_hints=0,dec_l0_once=0"; do
  NAME=${CFG%%:*}; SET=${CFG#*:}
  i=0
  for C in "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    (cd /tmp && MLD_BENCH_SET="$SET" timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcab_${NAME}_$i -o pmc -- \
      python $R/bench.py --profile-child --precision f16x3 --coalesce 32 --steps 1 > $R/gpurun_out/pmcab_${NAME}_$i.log 2>&1)
  done
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for name in ("on", "off"):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"gpurun_out/pmcab_{name}_*/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            e = agg[row["Kernel_Name"]][row["Counter_Name"]]; e[0] += 1; e[1] += float(row["Counter_Value"])
    res = {}
    for k, cs in agg.items():
        if not any(p in k for p in ("attn_flash", "strip_gemm_x3_kernel<6", "ffn_strip_x3", "den_loop")):
            continue
        ent = {c: v[1] / v[0] for c, v in cs.items()}
        ent["dispatches"] = max(v[0] for v in cs.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and "GRBM_GUI_ACTIVE" in ent:
            ent["mfma_busy_frac"] = round(ent["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (ent["GRBM_GUI_ACTIVE"] / 8), 4)
        if "SQ_INSTS_VALU" in ent and "SQ_INSTS_MFMA" in ent:
            ent["valu_per_mfma"] = round((ent["SQ_INSTS_VALU"] - ent["SQ_INSTS_MFMA"]) / ent["SQ_INSTS_MFMA"], 2)
        if "SQ_LDS_BANK_CONFLICT" in ent and "SQ_LDS_IDX_ACTIVE" in ent:
            ent["lds_conflict_frac"] = round(ent["SQ_LDS_BANK_CONFLICT"] / max(ent["SQ_LDS_IDX_ACTIVE"], 1.0), 4)
        res[k[:90]] = ent
    out[name] = res
json.dump({"note": "rocprofv3 --pmc, separate passes, bench.py --profile-child --coalesce 32 --steps 1; 'on' = defaults (attn_tr 1, nt_hints 1, dec_l0_once 1), "
           "'off' = all three 0.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1 024 SIMDs.", "configs": out},
          open("gpurun_out/r03c_pmc_sq_ab.json", "w"), indent=1)
for name, res in out.items():
    for k, e in res.items():
        print(name, k[:70], {c: e.get(c) for c in ("mfma_busy_frac", "valu_per_mfma", "lds_conflict_frac", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS")})
import os
for f in glob.glob("gpurun_out/pmcab_*/**/*.csv", recursive=True):
    if os.path.getsize(f) > 2 << 20: os.remove(f)
PY
