#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench (with the rocprofv3 summary its roofline block was computed from) +
# rocprofv3 kernel stats of the headline workload with 4 steps in flight.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
{
  echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" ; rocm-smi --showproductname 2>/dev/null | head -8
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
  echo "== bench"
  T0=$SECONDS
  MLD_BENCH_KEEP_ROCPROF=$PWD/gpurun_out/${TAG}_kernel_stats_bench_child.csv \
    timeout 600 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_${TAG}.err | tee gpurun_out/bench_${TAG}.json | cut -c1-3000
  echo "bench wall seconds: $((SECONDS - T0))"
  tail -5 gpurun_out/bench_${TAG}.err
} 2>&1 | tee gpurun_out/check_${TAG}.log
echo "== rocprofv3 (4 steps in flight)" | tee -a gpurun_out/check_${TAG}.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_inflight4 -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-alt --no-a2m --no-novae --no-clip --no-rocprof > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_inflight4.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_${TAG}_inflight4 -name "*kernel_trace*.csv" -size +20M -delete
f=$(find gpurun_out/prof_${TAG}_inflight4 -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_inflight4.csv
head -14 gpurun_out/${TAG}_kernel_stats_bench_child.csv | cut -c1-160 | tee -a gpurun_out/check_${TAG}.log
