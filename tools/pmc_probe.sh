export TMPDIR=/tmp; cd /tmp
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pmc_$T -o p -- python $GRAFT_REPO_ROOT/bench.py --profile-child --precision f16x3 --coalesce 32 --steps 1 > /tmp/pmc_$T.log 2>&1
  f=$(find /tmp/pmc_$T -name "*counter_collection*.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f:
    print("no file"); sys.exit()
a = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"][:60]
    if not any(s in k for s in ("den_loop", "ffn_strip", "attn_flash", "gemm_kernel<2, 4, 2, 2, false, true, 1, 8")):
        continue
    e = a[k][row["Counter_Name"]]; e[0] += 1; e[1] += float(row["Counter_Value"])
for k, cs in a.items():
    print(k, {c: round(v / n) for c, (n, v) in cs.items()}, "dispatches", max(n for n, _ in cs.values()))
PY
done
