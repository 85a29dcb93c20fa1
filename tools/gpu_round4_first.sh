#!/bin/bash
# The first gpurun call of the next round, prepared at the end of round 3 (whose GPU budget ran out before these could be measured):
#   0. build the alternative library HERE first:   make -C motion-latent-diffusion_amd/csrc alt HIPCC_EXTRA=-fno-slp-vectorize
#   1. tools/ab_build_flags.py       default build vs the no-SLP build, headline call shape
#   2. tools/ab_decode.py under rocprofv3 --stats: "ffn_swz" (swizzled LDS images in the decoder tail) and "final_strip" (decoder.norm +
#      final linear as one row-strip launch), interleaved; every variant is a kernel of its own name -> its own dispatch average
#   3. SQ counters of the loop with plain / swizzled images (LDS bank-conflict cycles: 1.43e9 per launch before the swizzle)
# ~3 GPU-minutes on a warm box.  Then flip the defaults that won, run tools/gpu_check.sh and tools/gpu_pmc.sh.
set -u
TAG=${1:-r04a}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$SECONDS
if [ -f motion-latent-diffusion_amd/mld_hip/libmldhip_alt.so ]; then
  timeout 300 python tools/ab_build_flags.py 2>&1 | tail -5 | tee gpurun_out/${TAG}_build_flags_ab.log | cut -c1-1200
else
  echo "no libmldhip_alt.so: step 1 skipped"
fi
AB='[{"ffn_swz": 0, "final_strip": 0}, {"ffn_swz": 1}, {"ffn_swz": 0}, {"ffn_swz": 1}, {"ffn_swz": 0, "final_strip": 1}, {"final_strip": 0}, {"final_strip": 1}, {"final_strip": 0}, {"ffn_swz": 1, "final_strip": 1}, {"ffn_swz": 0, "final_strip": 0}]'
(cd /tmp && AB_OPTS="$AB" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o ab -- python $R/tools/ab_decode.py > $R/gpurun_out/${TAG}_decoder_ab.log 2>&1)
grep -E "^\{" gpurun_out/${TAG}_decoder_ab.log | tail -1 | cut -c1-2000
F=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && { cp $F gpurun_out/${TAG}_kernel_stats_ab.csv; grep -E "Name|ffn_strip_x3|final_strip|layernorm_rows|gemm_kernel<2, 4, 2, 2, false, true, 1, 8|den_loop" $F | cut -c1-200; }
find gpurun_out/prof_${TAG} -name "*.csv" -size +4M -delete
for CFG in "swz:" "plain:fused_swz=0"; do
  NAME=${CFG%%:*}; SET=${CFG#*:}
  (cd /tmp && MLD_BENCH_SET="$SET" timeout 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
     -d $R/gpurun_out/pmc_${TAG}_$NAME -o pmc -- python $R/bench.py --profile-child --precision f16x3 --coalesce 32 --steps 1 > $R/gpurun_out/pmc_${TAG}_$NAME.log 2>&1)
  python - "$NAME" "$TAG" <<'PY'
import csv, glob, collections, sys
name, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(f"gpurun_out/pmc_{tag}_{name}/**/*counter_collection*.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "den_loop_kernel" in row["Kernel_Name"]:
            e = agg[row["Counter_Name"]]; e[0] += 1; e[1] += float(row["Counter_Value"])
print(name, {c: round(v[1] / v[0]) for c, v in agg.items()})
PY
done
echo "total seconds: $((SECONDS - T0))"
