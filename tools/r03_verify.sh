#!/bin/bash
# whole -m gpu suite + smoke + default bench line on one box
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu > gpurun_out/r03/gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r03/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail -5 gpurun_out/r03/bench_default.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03/bench_default.json"))
print(j["ms_per_step"], j["roofline"]["frac"], j["frame_check"], j.get("frame_check_kind"))
for k,v in j.get("other_configs",{}).items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_frame","frac","roofline")} if isinstance(v,dict) else v)
PY
