#!/bin/bash
# torch.distributed pipeline with split passes joined once per bucket by the caller
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp29; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split or gather or caller" > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 403 --warmup 30 --force-dist 1 --gather torch"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d["config"].get("frames_per_gather"), d["config"].get("split_streams"), d.get("frame_check"))'
for h in 136 544 1080; do for sp in 1 3; do
  $B --height $h --split-streams $sp 2>"$out/err.txt" | python3 -c "$P" "H=$h split=$sp" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done; done
$B --fast-math 0 2>"$out/err.txt" | python3 -c "$P" "H=1080 bit-exact default" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
