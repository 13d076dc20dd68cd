#!/bin/bash
mkdir -p gpurun_out/r03
python bench.py "$@" > gpurun_out/r03/bench_run.json 2> gpurun_out/r03/bench_run.err || { tail -8 gpurun_out/r03/bench_run.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03/bench_run.json"))
print(j["ms_per_step"], j["roofline"]["frac"], j["frame_check"])
print(json.dumps(j.get("frame_check_kind"), indent=0))
print(json.dumps(j["roofline"], indent=0))
for k,v in j.get("other_configs",{}).items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_frame","frac","roofline","ms_per_pass")} if isinstance(v,dict) else v)
PY
