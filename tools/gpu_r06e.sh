#!/bin/bash
# Round 6: the default bench (64 steps = two 2 048-motion calls) with its evidence file, for the record beside the driver's command.
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
export TMPDIR=/tmp
cd $repo
MLD_BENCH_EVIDENCE=$repo/gpurun_out/r06e_bench_evidence.json MLD_BENCH_KEEP_ROCPROF=$repo/gpurun_out/r06e_kernel_stats_bench_child.csv \
  timeout 1500 python bench.py 2>gpurun_out/r06e_bench.err | tee gpurun_out/r06e_bench.json | cut -c1-1500
grep -v "^EVIDENCE" gpurun_out/r06e_bench.err | tail -3
