set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; cd /tmp
for mode in pipelined serial; do
  rm -rf /tmp/prof_$mode
  if [ $mode = serial ]; then export TRACE_SERIAL=1; else unset TRACE_SERIAL; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$mode -o t -- python $repo/tools/trace_pipeline.py > /tmp/tp_$mode.log 2>&1
  f=$(find /tmp/prof_$mode -name "*kernel_trace.csv" | head -1)
  echo -n "$mode "; python $repo/tools/trace_pipeline.py --parse $f
done 2>&1 | tee $repo/gpurun_out/r06_trace_pipeline.log
