#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp27; mkdir -p "$out"
export TMPDIR=/tmp
echo "== torch pipeline host cost"; timeout -k 5 300 python3 tools/torch_pipeline_probe.py 2>"$out/torch.err" | tee "$out/torch_pipeline.json"; tail -2 "$out/torch.err"
echo "== shard8 probe"; timeout -k 5 400 python3 tools/shard8_probe.py "$out/r02_shard8.json" > "$out/shard8.log" 2>&1; tail -22 "$out/shard8.log"
