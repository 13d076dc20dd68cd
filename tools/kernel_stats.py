#!/usr/bin/env python3
"""Per-kernel statistics of a `rocprofv3 --kernel-trace --stats -d <dir> -o kt -- <command>` run, as CSV on stdout:
    python tools/kernel_stats.py <dir> ["the command that was profiled"]
Reads the rocpd SQLite database rocprofv3 writes (or its *kernel_stats.csv, depending on version)."""
import csv
import glob
import os
import sqlite3
import sys

out = sys.argv[1]
if len(sys.argv) > 2:
    print("# " + sys.argv[2])
print("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage")
stats = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
if stats:
    for r in csv.DictReader(open(stats[0])):
        print('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.3f' % (r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                   float(r.get("MinNs", 0)) / 1e3, float(r.get("MaxNs", 0)) / 1e3, float(r["Percentage"])))
elif dbs:
    db = sqlite3.connect(dbs[0])
    lo_hi = {n: (lo, hi) for n, lo, hi in db.execute("select name, min(duration), max(duration) from kernels group by name")}
    for n, c, t, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        lo, hi = lo_hi.get(n, (0, 0))
        print('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.3f' % (n, c, t, avg, lo / 1e3, hi / 1e3, pct))     # the view is in us, kernels.duration in ns
else:
    sys.exit("no rocprofv3 kernel statistics under %s" % out)
