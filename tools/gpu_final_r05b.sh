set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
} 2>&1 | tee gpurun_out/check_r05b.log
T0=$SECONDS
MLD_BENCH_KEEP_ROCPROF=$PWD/gpurun_out/r05b_kernel_stats_bench_child_s20.csv MLD_BENCH_KEEP_ROCPROF_SINGLE=$PWD/gpurun_out/r05b_kernel_stats_single_request.csv \
  timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/bench_r05b_s20.err > gpurun_out/bench_r05b_s20.json
echo "driver command wall seconds: $((SECONDS - T0))" | tee -a gpurun_out/check_r05b.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r05b_s20.json").read().strip().splitlines()[-1])
print(d["value"], d["value_single_batch"], d["ms_per_step_single_batch"], d["roofline"]["frac"], d["single_batch"]["roofline"].get("traffic"), d["single_batch"]["roofline"].get("traffic_source"))
PY
