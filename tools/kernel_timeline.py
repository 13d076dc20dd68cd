#!/usr/bin/env python3
"""Timeline of the last dispatches of a `rocprofv3 --kernel-trace -d <dir>` run (rocpd SQLite database):
    python tools/kernel_timeline.py <dir> [n_last=24] [name-substring]
Prints start (us, relative), duration and the name of each dispatch, then per kernel name the average duration and the busy
fraction of the traced span — how the launches of a frame overlap on the chip."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
want = sys.argv[3] if len(sys.argv) > 3 else "k_mcm"
dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
if not dbs:
    sys.exit("no rocpd database under %s" % out)
db = sqlite3.connect(dbs[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in db.execute("select * from kernels order by start")]
rows = [r for r in rows if want in str(r.get("name", ""))]
if not rows:
    sys.exit("no dispatch matching %r; columns: %s" % (want, cols))
tail = rows[-n_last:]
t0 = tail[0]["start"]
for r in tail:
    print("%10.2f us  +%8.2f us  grid %-8s %s" % ((r["start"] - t0) / 1e3, (r["end"] - r["start"]) / 1e3, r.get("grid_x", r.get("grid_size", "?")), str(r["name"])[:90]))
half = rows[len(rows) // 2:]
span = (max(r["end"] for r in half) - min(r["start"] for r in half)) / 1e3
by = {}
for r in half:
    k = (str(r["name"])[:70], r.get("grid_x", r.get("grid_size", "?")))
    by.setdefault(k, []).append((r["end"] - r["start"]) / 1e3)
print("second half of the trace: %.1f us, %d dispatches" % (span, len(half)))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("  %-70s grid %-8s n %5d avg %8.2f us  sum/span %.3f" % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / span))
