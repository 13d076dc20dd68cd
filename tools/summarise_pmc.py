#!/usr/bin/env python3
"""Condenses one tools/pmc.sh directory: kernel_stats.csv (rocprofv3 --kernel-trace --stats) and summary.json with, PER KERNEL of
the renderer (k_mcm_integrate, k_mcm_miss, k_eam, ...), the mean per-launch counter values, corrected as MI355X_MICROARCH.md
prescribes (gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes -> doubled; WRITE_SIZE exact; both in KiB)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
summary = {"_command": open(os.path.join(d, "command.txt")).read().strip(), "kernels": {}}
# the traced run's own bench line (kt.log holds bench.py's stdout): the tracer's overhead is on record next to the kernel durations
traced = None
try:
    for line in open(os.path.join(d, "kt.log"), errors="replace"):
        if line.startswith('{"metric"'):
            j = json.loads(line)
            traced = {"ms_per_step": j["ms_per_step"], "steps": j["steps"], "value": j["value"]}
except OSError:
    pass
summary["_traced_run"] = traced
stats = glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(d, "kernel_stats.csv"), "w") as f:
        f.write("# %s\n# the traced run's own figure: ms_per_step %s over %s steps (rocprofv3 --kernel-trace active, no time-based warm-up)\nName,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage\n"
                % (summary["_command"], ("%.5f" % traced["ms_per_step"]) if traced else "?", traced["steps"] if traced else "?"))
        for r in rows[:12]:
            f.write('"%s",%s,%.3f,%.3f,%.3f,%.3f,%s\n' % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                         float(r.get("MinNs", 0)) / 1e3, float(r.get("MaxNs", 0)) / 1e3, r["Percentage"]))
    for r in rows:
        if r["Name"].startswith("void k_") and int(r["Calls"]) >= 50:
            summary["kernels"].setdefault(r["Name"][:100], {})["kernel_stats"] = {
                "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r.get("MinNs", 0)) / 1e3, "max_us": float(r.get("MaxNs", 0)) / 1e3}
acc = {}
for path in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:100]
        if k in summary["kernels"]:
            a = acc.setdefault((k, r["Counter_Name"]), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    summary["kernels"][k].setdefault("counters_mean_per_launch", {})[c] = s / n
for k, e in summary["kernels"].items():
    m = e.get("counters_mean_per_launch", {})
    der = {}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        der["hbm_traffic_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
        der["valu_instructions_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    if "VALUBusy" in m:
        der["valu_busy_frac"] = m["VALUBusy"] / 100.0
    if "TA_BUSY_avr" in m and m.get("GRBM_GUI_ACTIVE"):
        der["ta_busy_frac"] = m["TA_BUSY_avr"] / (m["GRBM_GUI_ACTIVE"] / 8.0)
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in m:
                der[c.lower() + "_per_wave_cycle"] = m[c] / m["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in m and (m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0)):
        der["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    e["_derived"] = der
    print(k[:60], json.dumps(e.get("kernel_stats", {})), json.dumps(der))
json.dump(summary, open(os.path.join(d, "summary.json"), "w"), indent=1)
