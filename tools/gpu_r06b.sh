#!/bin/bash
# Round 6, second GPU call: cluster-loop changes (entry check, host status word, per-XCD capacity), the pipelined sample_many, the foreign-stream test; then the driver's bench command.
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
export TMPDIR=/tmp
cd $repo
{
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -4
  echo "== pytest (cluster loop, pipelined, foreign stream)"
  timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "pipelined or foreign or cluster or native_library or graph_replay" 2>&1 | tail -25
  echo "== bench --steps 20 --warmup 5"
  T0=$SECONDS
  MLD_BENCH_EVIDENCE=$repo/gpurun_out/r06b_bench_evidence_s20.json MLD_BENCH_KEEP_ROCPROF=$repo/gpurun_out/r06b_kernel_stats_bench_child_s20.csv MLD_BENCH_KEEP_ROCPROF_SINGLE=$repo/gpurun_out/r06b_kernel_stats_single_request.csv \
    timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r06b_bench.err | tee gpurun_out/r06b_bench_s20.json | cut -c1-6000
  echo "bench wall seconds: $((SECONDS - T0)); line bytes: $(wc -c < gpurun_out/r06b_bench_s20.json)"
  grep -v "^EVIDENCE" gpurun_out/r06b_bench.err | tail -5
} 2>&1 | tee $repo/gpurun_out/r06b.log
