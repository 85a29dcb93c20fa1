#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench (with the rocprofv3 summaries its roofline blocks were computed from).
# Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r03}
{
  echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" ; rocm-smi --showproductname 2>/dev/null | head -8
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
  echo "== bench"
  T0=$SECONDS
  MLD_BENCH_KEEP_ROCPROF=$PWD/gpurun_out/${TAG}_kernel_stats_bench_child.csv MLD_BENCH_KEEP_ROCPROF_SINGLE=$PWD/gpurun_out/${TAG}_kernel_stats_single_request.csv \
    timeout 900 python bench.py 2>gpurun_out/bench_${TAG}.err | tee gpurun_out/bench_${TAG}.json | cut -c1-3000
  echo "bench wall seconds: $((SECONDS - T0))"
  tail -5 gpurun_out/bench_${TAG}.err
} 2>&1 | tee gpurun_out/check_${TAG}.log
head -14 gpurun_out/${TAG}_kernel_stats_bench_child.csv | cut -c1-160 | tee -a gpurun_out/check_${TAG}.log
