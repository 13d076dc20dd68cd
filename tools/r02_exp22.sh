#!/bin/bash
# brickify in 16x4x4 blocks (parity + rate), the caller-owned render target test in its worker process, host-bound regimes of the
# gather pipeline at a shard's frame size
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp22; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_readers.py tests/test_gpu_fuzz.py -x -q -m gpu > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
echo "== brickify"; timeout -k 5 300 python3 tools/brickify_rate.py > "$out/brickify.json" 2>"$out/brickify.err"; cat "$out/brickify.json"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 400 --warmup 30 --check 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("split_streams"), d["config"].get("frames_per_launch"))'
for h in 136 272 544; do
for cfg in "--force-dist 0 --split-streams 1" "--force-dist 0 --split-streams 1 --frames-per-launch 16 --graph 0" "--force-dist 1 --gather native --split-streams 1" "--force-dist 1 --gather native --split-streams 1 --frames-per-launch 16" "--force-dist 1 --gather native --split-streams 2" "--force-dist 1 --gather native --split-streams 2 --frames-per-launch 16" "--force-dist 1 --gather native --split-streams 3 --frames-per-launch 16"; do
  $B --height $h $cfg 2>"$out/err.txt" | python3 -c "$P" "H=$h $cfg" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done; done
