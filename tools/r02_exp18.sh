#!/bin/bash
# per-frame cost of the N > 1 pipelines on one rank (world = 1, RCCL initialised): host-bound or not?
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp18; mkdir -p "$out"
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("parallelism"), d["config"].get("gather_calibration"), d.get("frame_check"))'
for cfg in "--force-dist 0" "--force-dist 1 --gather torch" "--force-dist 1 --gather native" "--force-dist 1 --gather native --frames-per-launch 16"; do
  $B $cfg 2>"$out/err.txt" | python3 -c "$P" "$cfg" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done
