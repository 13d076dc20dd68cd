#!/bin/bash
# brickify strip rewrite (parity + rate), caller-owned render target with split streams, the N > 1 pipelines at world = 1
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp19; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_readers.py tests/test_gpu_fuzz.py tests/test_gpu_fast_math.py -x -q -m gpu > "$out/parity.txt" 2>&1; tail -3 "$out/parity.txt"
echo "== brickify"; timeout -k 5 300 python3 tools/brickify_rate.py > "$out/brickify.json" 2>"$out/brickify.err"; cat "$out/brickify.json"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("parallelism"), d["config"].get("gather_calibration"), d.get("frame_check"))'
for cfg in "--force-dist 1 --gather torch" "--force-dist 1 --gather native" "--force-dist 1 --gather native --fast-math 0"; do
  $B $cfg 2>"$out/err.txt" | python3 -c "$P" "$cfg" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done
