#!/bin/bash
# One gpurun call: smoke + GPU parity tests + bench + rocprofv3 kernel stats.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
{
  echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" ; rocm-smi --showproductname 2>/dev/null | head -8
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
  echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/bench_${TAG}.err | tee gpurun_out/bench_${TAG}.json
  tail -5 gpurun_out/bench_${TAG}.err
} 2>&1 | tee gpurun_out/check_${TAG}.log
echo "== rocprofv3" | tee -a gpurun_out/check_${TAG}.log
# Kernel stats of the HEADLINE workload only, one step in flight: the per-kernel averages then correspond to the isolated
# per-kernel timings bench.py reports in `kernels` / `roofline` (with 4 steps in flight kernels of different batches
# overlap and each one's wall duration is longer; the secondary workloads reuse the same kernels at other shapes).
prof() {  # $1 = suffix, rest = bench flags
  local sfx=$1; shift
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${sfx} -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-a2m --no-novae --no-clip "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${sfx}.log 2>&1
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof_${TAG}_${sfx} -name "*kernel_trace*.csv" -size +20M -delete
}
prof inflight1 --in-flight 1
prof inflight4 --in-flight 4
f=$(find gpurun_out/prof_${TAG}_inflight1 -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -24 "$f" | tee -a gpurun_out/check_${TAG}.log
tail -3 gpurun_out/prof_${TAG}_inflight1.log
