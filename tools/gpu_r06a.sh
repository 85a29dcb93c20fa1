#!/bin/bash
# Round 6, first GPU call: the opt-in half-Q|K|V decoder form -- GPU parity test, A/B of decode time and error (tools/ab_decode_half.py), rocprofv3 per-kernel
# averages of the same A/B (in-projection / attention kernels of both forms side by side).  Everything lands in gpurun_out/.
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
export TMPDIR=/tmp
cd $repo
{
  echo "== pytest (decoder forms)"
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "decoder_half or split_f16_decode_mode or first_decoder_layer or native_library" 2>&1 | tail -15
  echo "== ab_decode_half"
  timeout 900 python tools/ab_decode_half.py 2>&1 | tee gpurun_out/r06_decoder_half_ab.log | tail -12
  tail -1 gpurun_out/r06_decoder_half_ab.log > gpurun_out/r06_decoder_half_ab.json
  echo "== rocprofv3 of the A/B"
  cd /tmp && rm -rf /tmp/prof_ab
  AB_TIMING_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o ab -- python $repo/tools/ab_decode_half.py > /tmp/ab_prof.log 2>&1
  f=$(find /tmp/prof_ab -name "*kernel_stats.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp "$f" $repo/gpurun_out/r06_kernel_stats_decoder_half_ab.csv; head -30 "$f" | cut -c1-200; else echo "no stats file"; tail -5 /tmp/ab_prof.log; fi
} 2>&1 | tee $repo/gpurun_out/r06a.log
