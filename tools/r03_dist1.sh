#!/bin/bash
# the N > 1 code paths of bench.py on ONE rank (one-rank RCCL group): torch.distributed pipeline, then torch-first + native
mkdir -p gpurun_out/r03
for args in "--gather auto --force-dist 2" "--gather torch --force-dist 1" "--gather native --safe-first 2 --force-dist 1"; do
  timeout -k 10 400 python bench.py --other-configs 0 --cpu-baseline 0 --stream-probe 0 $args > gpurun_out/r03/dist1.json 2> gpurun_out/r03/dist1.err; rc=$?
  echo "== $args (rc $rc)"
  [ $rc -ne 0 ] && tail -5 gpurun_out/r03/dist1.err
  python3 - <<'PY'
import json
try:
    j=json.load(open("gpurun_out/r03/dist1.json"))
    print(j["ms_per_step"], j["frame_check"], j["config"]["parallelism"], j["config"].get("other_pipeline_ms_per_step"), j["config"]["split_streams"], j["config"].get("native_preflight"))
    b = j["config"].get("bucket_kernel_form")
    if b: print("   bucket kernel form:", b["ms_per_step"], b["frame_check"], b["bucket_launches"], "frames per launch", b["frames_per_launch"])
    d = j["config"].get("display_gather_form")
    if d: print("   display gather form:", d["ms_per_step"], d["frame_check"], d["bucket_launches"], d["bytes_per_frame_and_xgmi_link"])
except Exception as e:
    print("no line:", e)
PY
done
