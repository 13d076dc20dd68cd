#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into small JSON/CSV summaries (printed, and written next to
the raw data as summary_<tag>.json / kernel_stats_<tag>.csv; copy those into profiles/)."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
KERNEL = "k_mcm_integrate"
summary = {"_kernel_filter": KERNEL, "_command": "python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 100 --warmup 10"}

import sqlite3

# rocprofv3 writes either CSV files or one rocpd SQLite database per pass, depending on its version / -o flag
stats = glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True)
dbs = glob.glob(os.path.join(out, "kt", "**", "*_results.db"), recursive=True)
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
elif dbs:
    rows = list(sqlite3.connect(dbs[0]).execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lo_hi = {n: (lo, hi) for n, lo, hi in sqlite3.connect(dbs[0]).execute("select name, min(duration), max(duration) from kernels group by name")}
if rows:
    with open(os.path.join(out, "kernel_stats_%s.csv" % tag), "w") as f:
        f.write("Name,Calls,TotalDurationUs,AverageUs,Percentage\n")
        for n, c, t, avg, pct in rows:
            f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (n, c, t, avg, pct))
    for n, c, t, avg, pct in rows:
        if KERNEL in n:
            summary["kernel_stats"] = {"name": n[:80], "calls": int(c), "avg_us": float(avg), "percent": float(pct)}
counters = {}
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        c = counters.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*_results.db"), recursive=True):
    q = "select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by counter_name"
    for name, total, n in sqlite3.connect(path).execute(q, ("%" + KERNEL + "%",)):
        c = counters.setdefault(name, [0.0, 0])
        c[0] += float(total); c[1] += int(n)
summary["counters_mean_per_launch"] = {k: {"mean": v[0] / v[1], "launches": v[1]} for k, v in sorted(counters.items())}
m = {k: v["mean"] for k, v in summary["counters_mean_per_launch"].items()}
d = {}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2
    d["hbm_traffic_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
    d["valu_instructions_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
    d["valu_lane_utilisation"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
if "SQ_WAVE_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    d["avg_resident_waves_per_cu"] = m["SQ_WAVE_CYCLES"] / m["GRBM_GUI_ACTIVE"] / 8.0      # as derived in round r01: 8 XCD-level counter instances
summary["_derived"] = d
s = json.dumps(summary, indent=1)
open(os.path.join(out, "summary_%s.json" % tag), "w").write(s + "\n")
print(s)
