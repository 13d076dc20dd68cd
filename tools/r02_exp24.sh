#!/bin/bash
# the gather's "rendered" events attached to the dispatches (hipExtLaunchKernel stop event) vs hipEventRecord behind them
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp24; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gather or split or shard" > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 400 --warmup 30 --gather-root 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("split_streams"), d.get("frame_check"))'
for h in 136 272 544 1080; do
$B --height $h --force-dist 0 --split-streams 1 --check 0 2>/dev/null | python3 -c "$P" "H=$h plain" | tee -a "$out/ab.txt"
for x in 0 1; do for sp in 1 3; do
  VPT_GATHER_EVENTS=$x $B --height $h --force-dist 1 --gather native --split-streams $sp 2>"$out/err.txt" | python3 -c "$P" "H=$h gather ride=$x split=$sp" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done; done; done
