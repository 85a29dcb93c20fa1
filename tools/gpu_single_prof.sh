#!/bin/bash
# rocprofv3 kernel summary of the single-request path (one bs-64 batch at a time, latency kernels): tools/gpu_single_prof.sh <tag> [opt:val ...]
# -> gpurun_out/<tag>_kernel_stats_single.csv
set -u
tag=${1:-r03}; shift || true
repo=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_single
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single -o single -- python $repo/tools/run_single.py 4 "$@" > /tmp/rs.log 2>&1
grep -v rocprofv3 /tmp/rs.log | tail -4
f=$(find /tmp/prof_single -name "*kernel_stats.csv" 2>/dev/null | head -1)
t=$(find /tmp/prof_single -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$t" ]; then python3 - "$t" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last batch: the final 2098 dispatches or so; report durations and gaps of the loop kernels in it
rows = rows[-2000:]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
import statistics as st
print("last 2000 dispatches: span %.3f ms, sum of durations %.3f ms, sum of gaps %.3f ms; median duration %.2f us, median gap %.2f us, p90 gap %.2f us" % (
    span / 1e6, sum(dur) / 1e6, sum(gap) / 1e6, st.median(dur) / 1e3, st.median(gap) / 1e3, sorted(gap)[int(len(gap) * 0.9)] / 1e3))
PY
fi
if [ -n "$f" ]; then cp "$f" $repo/gpurun_out/${tag}_kernel_stats_single.csv; head -22 "$f" | cut -c1-220; else echo "no stats file"; fi
