#!/bin/bash
# Last call of round 3 (2.4 GPU-minutes left): the loop kernel's PMC traffic passes on the final build (the swizzled loop is a new
# kernel: bench.py refuses the previous summary), then -- if the budget lasts -- the GPU test of the benchmarked serving shape.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
PMC_TRAFFIC_ONLY=1 PMC_TIMEOUT=60 bash tools/gpu_pmc.sh r03 > gpurun_out/r03_pmc.log 2>&1
python -c "import json; d=json.load(open('gpurun_out/r03_pmc_traffic.json')); print(d['loop_kernel_code_hash'], d['kernels']['den_loop'])"
timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "headline_serving_shape" 2>&1 | tail -3 | tee gpurun_out/r03d_headline_test.log
