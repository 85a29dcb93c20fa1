#!/bin/bash
# One gpurun call that re-stamps round 5's evidence on the final build: smoke, pytest -m gpu, the driver's command (bench.py --gpus 1 --steps 20 --warmup 5)
# with its rocprofv3 children (the headline call AND the single bs-64 request), the default bench.py, the traffic PMC passes at 1 / 20 requests per call,
# the cluster-loop A/B (tools/ab_cluster.py) and the harness phase stamps.
#   tools/gpu_final_r05.sh [TAG=r05]     -> gpurun_out/<TAG>_*  (copy what is cited into profiles/)
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)"; rocm-smi --showproductname 2>/dev/null | head -6
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -4
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
} 2>&1 | tee gpurun_out/check_${TAG}.log
T0=$SECONDS
MLD_BENCH_KEEP_ROCPROF=$PWD/gpurun_out/${TAG}_kernel_stats_bench_child_s20.csv MLD_BENCH_KEEP_ROCPROF_SINGLE=$PWD/gpurun_out/${TAG}_kernel_stats_single_request.csv \
  timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_${TAG}_s20.err > gpurun_out/bench_${TAG}_s20.json
echo "driver command wall seconds: $((SECONDS - T0))" | tee -a gpurun_out/check_${TAG}.log
cut -c1-400 gpurun_out/bench_${TAG}_s20.json; tail -3 gpurun_out/bench_${TAG}_s20.err
if [ "${FINAL_DEFAULT_BENCH:-1}" = 1 ]; then
  T0=$SECONDS
  MLD_BENCH_KEEP_ROCPROF=$PWD/gpurun_out/${TAG}_kernel_stats_bench_child.csv timeout 700 python bench.py --no-alt --no-a2m --no-novae --no-clip 2>gpurun_out/bench_${TAG}.err > gpurun_out/bench_${TAG}.json
  echo "default bench (headline blocks only) wall seconds: $((SECONDS - T0))" | tee -a gpurun_out/check_${TAG}.log
  cut -c1-300 gpurun_out/bench_${TAG}.json
fi
PMC_TRAFFIC_ONLY=${PMC_TRAFFIC_ONLY:-1} PMC_SHAPES="${PMC_SHAPES:-1 20}" PMC_TIMEOUT=150 bash tools/gpu_pmc.sh $TAG 2>&1 | tail -4
timeout 250 python tools/ab_cluster.py --batches 64,128 --rounds 5 --variants x3_launches,x3_cluster_wt,x3_cluster_plain,x3_cluster_4groups --oracle 1 --out gpurun_out/${TAG}_cluster_ab.json 2>&1 | grep -E "^(64|128) " | cut -c1-240
[ -x build/lb/cluster_cg8_trace ] && timeout 60 build/lb/cluster_cg8_trace 64 3 50 2>&1 | grep -v checksums | tee gpurun_out/${TAG}_cluster_trace.json | cut -c1-300
