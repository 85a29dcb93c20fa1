#!/bin/bash
# Quick GPU iteration: parity tests + phase trace + short bench (no CPU baseline, no rocprof).
set -u
mkdir -p gpurun_out
TAG=${1:-quick}
{
  echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6
  echo "== trace"; timeout 300 python tools/trace_tile32.py 2>&1 | grep -v amdgpu.ids
  echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_${TAG}.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
} 2>&1 | tee gpurun_out/quick_${TAG}.log
