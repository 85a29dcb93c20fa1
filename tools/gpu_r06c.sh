#!/bin/bash
# Round 6: the whole GPU suite + the driver's bench command (evidence + rocprofv3 summaries kept).
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
export TMPDIR=/tmp
cd $repo
TAG=${1:-r06c}
{
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
  echo "== pytest -m gpu"
  timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -60
  echo "== bench --steps 20 --warmup 5"
  T0=$SECONDS
  MLD_BENCH_EVIDENCE=$repo/gpurun_out/${TAG}_bench_evidence_s20.json MLD_BENCH_KEEP_ROCPROF=$repo/gpurun_out/${TAG}_kernel_stats_bench_child_s20.csv MLD_BENCH_KEEP_ROCPROF_SINGLE=$repo/gpurun_out/${TAG}_kernel_stats_single_request.csv \
    timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench_s20.json | cut -c1-3000
  echo "bench wall seconds: $((SECONDS - T0)); line bytes: $(wc -c < gpurun_out/${TAG}_bench_s20.json)"
  grep -v "^EVIDENCE" gpurun_out/${TAG}_bench.err | tail -5
} 2>&1 | tee $repo/gpurun_out/${TAG}.log
