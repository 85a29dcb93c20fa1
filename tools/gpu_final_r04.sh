#!/bin/bash
# One gpurun call that re-stamps the round's evidence on the final build: tools/gpu_check.sh (smoke, pytest -m gpu, default bench with its
# rocprofv3 children), the driver's command (bench.py --gpus 1 --steps 20 --warmup 5) with its own children, the PMC passes at both call
# shapes, and a rocprofv3 kernel summary of the diffusion-only variant (config 4, split-f16, 40 DDPM steps, tools/ab_novae_gemm.py).
#   tools/gpu_final_r04.sh [TAG=r04b]     -> gpurun_out/<TAG>_*  (copy what is cited into profiles/)
set -u
TAG=${1:-r04b}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_check.sh $TAG
MLD_BENCH_KEEP_ROCPROF=$PWD/gpurun_out/${TAG}_kernel_stats_bench_child_s20.csv \
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_${TAG}_s20.err > gpurun_out/bench_${TAG}_s20.json
cut -c1-600 gpurun_out/bench_${TAG}_s20.json
PMC_TIMEOUT=150 bash tools/gpu_pmc.sh $TAG 2>&1 | tail -4
( cd /tmp && rm -rf /tmp/prof_novae && NOVAE_STEPS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_novae -o nv -- \
    python $GRAFT_REPO_ROOT/tools/ab_novae_gemm.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_novae_prof.log 2>&1 )
f=$(find /tmp/prof_novae -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_novae_f16x3.csv && head -12 "$f" | cut -c1-170
