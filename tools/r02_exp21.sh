#!/bin/bash
# split ranges inside the native gather pipeline: parity, then per-frame time at world = 1 (full frame) and as rank 3 of 8's frame size
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp21; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
echo "== import order"; timeout -k 5 200 python3 /dev/stdin > "$out/order.txt" 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import vpt_amd
ctx = vpt_amd.Context(0); ctx.synchronize()
import torch
x = torch.zeros(1024, device='cuda'); torch.cuda.synchronize()
print("library first, torch second: ran", float(x.sum()))
PY
echo "exit $?"; tail -1 "$out/order.txt"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("parallelism")[:60], d["config"].get("split_streams"), d.get("frame_check"))'
for cfg in "--force-dist 1 --gather native --split-streams 1" "--force-dist 1 --gather native --split-streams 3" "--force-dist 1 --gather native --split-streams 3 --fast-math 0" "--force-dist 1 --gather torch --split-streams 3" "--force-dist 1 --gather native --split-streams 3 --height 136" "--force-dist 1 --gather native --split-streams 1 --height 136" "--force-dist 0 --split-streams 3 --height 136 --check 0" "--force-dist 0 --split-streams 1 --height 136 --check 0"; do
  $B $cfg 2>"$out/err.txt" | python3 -c "$P" "$cfg" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done
