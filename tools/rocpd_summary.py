#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table (markdown/CSV-ish)."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
    f.write("# source db: %s\n" % db)
    f.write("name,calls,total_us,avg_us,percent\n")
    for n, calls, tot, avg, pct in rows:
        f.write('"%s",%d,%.1f,%.3f,%.2f\n' % (n, calls, tot / 1e3 if tot > 1e7 else tot, avg / 1e3 if avg > 1e4 else avg, pct))
print(open(out).read())
