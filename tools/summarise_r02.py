#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/r02_profiles.sh into small files under <dir>/summary/ (copy those into profiles/):
per configuration the kernel statistics CSV (rocprofv3 --kernel-trace --stats) and a JSON with the mean per-launch counter values of
the dominant kernel, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950: doubled;
WRITE_SIZE exact; both in KiB)."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
os.makedirs(os.path.join(out, "summary"), exist_ok=True)
FILTER = {"mip256": "k_mip", "eam256": "k_eam"}
for d in sorted(glob.glob(os.path.join(out, "*", ""))):
    name = os.path.basename(os.path.dirname(d))
    if name == "summary" or not os.path.exists(os.path.join(d, "command.txt")):
        continue
    kfilter = FILTER.get(name, "k_mcm_integrate")
    summary = {"_command": open(os.path.join(d, "command.txt")).read().strip(), "_kernel_filter": kfilter}
    stats = glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out, "summary", "r02_%s_kernel_stats.csv" % name), "w") as f:
            f.write("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage\n")
            for r in rows[:12]:
                f.write('"%s",%s,%.3f,%.3f,%.3f,%.3f,%s\n' % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                             float(r.get("MinNs", 0)) / 1e3, float(r.get("MaxNs", 0)) / 1e3, r["Percentage"]))
        for r in rows:
            if kfilter in r["Name"]:
                summary["kernel_stats"] = {"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                           "min_us": float(r.get("MinNs", 0)) / 1e3, "max_us": float(r.get("MaxNs", 0)) / 1e3, "percent": float(r["Percentage"])}
                break
    acc = {}
    for path in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kfilter in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    m = {k: v[0] / v[1] for k, v in acc.items()}
    summary["counters_mean_per_launch"] = {k: {"mean": m[k], "launches": acc[k][1]} for k in sorted(m)}
    der = {}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        der["hbm_traffic_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
        der["valu_instructions_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    if "VALUBusy" in m:
        der["valu_busy_frac"] = m["VALUBusy"] / 100.0
    if "TA_BUSY_avr" in m and "GRBM_GUI_ACTIVE" in m:
        der["ta_busy_frac"] = m["TA_BUSY_avr"] / (m["GRBM_GUI_ACTIVE"] / 8.0)
    summary["_derived"] = der
    json.dump(summary, open(os.path.join(out, "summary", "r02_%s_pmc.json" % name), "w"), indent=1)
    print(name, json.dumps(summary.get("kernel_stats", {})), json.dumps(der))
