#!/bin/bash
# Round 6: the split-graph schedule of "many_pipeline" -- its tests, the kernel-trace timeline, and the bench line.
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
export TMPDIR=/tmp
cd $repo
{
  echo "== pytest pipelined / sample_many / cluster"
  timeout 1500 python -m pytest tests -m gpu -q -s -k "pipelined or sample_many or cluster" 2>&1 | grep -v "^$" | tail -30
  echo "== trace"
  bash tools/gpu_trace_pipeline.sh 2>&1 | tail -40
  echo "== bench --steps 20 --warmup 5"
  MLD_BENCH_EVIDENCE=$repo/gpurun_out/r06l_bench_evidence_s20.json timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r06l_bench.err | tee gpurun_out/r06l_bench_s20.json | cut -c1-2500
} 2>&1 | tee $repo/gpurun_out/r06l.log
