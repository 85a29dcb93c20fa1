#!/bin/bash
# Round-end evidence in one gpurun call (before tools/gpu_check.sh, whose bench reads the PMC summary this writes):
#   PMC passes (traffic + SQ counters) at the headline call shape, the loop's phase counters, the latency kernels' phase trace,
#   loop / decoder / single-batch option A/Bs, the engine-side block of the precision attribution.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
PMC_TIMEOUT=150 bash tools/gpu_pmc.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
timeout 200 python tools/trace_loop.py > gpurun_out/${TAG}_trace_loop.log 2>&1
timeout 120 python tools/trace_tile32.py > gpurun_out/${TAG}_trace_tile32.log 2>&1
timeout 200 python tools/ab_fused_opts.py > gpurun_out/${TAG}_loop_ab.log 2>&1
# (round 4: the first entries measure "ffn_swz", built after round 3's GPU budget was spent)
AB_OPTS='[{"ffn_swz": 0}, {"ffn_swz": 1}, {"ffn_swz": 0}, {"ffn_swz": 1}, {"ffn_swz": 0}, {"ffn_strip": 6, "dec_tail": 0}, {"ffn_strip": 3, "dec_tail": 0}, {"ffn_strip": 3, "dec_tail": 1}, {"ffn_strip": 6, "dec_tail": 0}, {"ffn_strip": 3, "dec_tail": 0}, {"ffn_strip": 3, "dec_tail": 1}, {"ffn_strip": 1, "dec_tail": 1, "strip_ring": 4}, {"strip_ring": 8, "strip_gemm": 0}, {"strip_gemm": 1, "ffn_strip": 0}, {"ffn_strip": 1, "flash_attn": 0}, {"flash_attn": 1}]' \
  timeout 300 python tools/ab_decode.py > gpurun_out/${TAG}_decoder_ab.log 2>&1
timeout 150 python tools/ab_single.py --out gpurun_out/${TAG}_single_batch_ab.json > gpurun_out/${TAG}_single_ab.log 2>&1
timeout 200 python tools/precision_attribution.py --gpu-only --out profiles/r03_precision_ab.json > gpurun_out/${TAG}_precision_gpu.log 2>&1
cp profiles/r03_precision_ab.json gpurun_out/${TAG}_precision_ab.json
tail -3 gpurun_out/${TAG}_pmc.log | cut -c1-600
tail -1 gpurun_out/${TAG}_trace_loop.log | cut -c1-400
tail -2 gpurun_out/${TAG}_loop_ab.log | cut -c1-600
tail -1 gpurun_out/${TAG}_decoder_ab.log | cut -c1-1500
tail -3 gpurun_out/${TAG}_single_ab.log | cut -c1-300
tail -2 gpurun_out/${TAG}_precision_gpu.log | cut -c1-600
