#!/bin/bash
# the torch.distributed pipeline with bucketed all_gathers (--frames-per-gather) on a one-rank RCCL group, frames of a shard's size
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp28; mkdir -p "$out"
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 403 --warmup 30 --force-dist 1 --gather torch"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("frames_per_gather"), d.get("frame_check"))'
for h in 136 272 544 1080; do for f in 1 4 8; do
  $B --height $h --frames-per-gather $f 2>"$out/err.txt" | python3 -c "$P" "H=$h F=$f" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done; done
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
echo "== shard8 probe"; timeout -k 5 400 python3 tools/shard8_probe.py "$out/r02_shard8.json" > "$out/shard8.log" 2>&1; tail -8 "$out/shard8.log"
