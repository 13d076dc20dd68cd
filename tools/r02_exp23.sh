#!/bin/bash
# where the gather pipeline's 8 us per frame go (timing-only experiment library): bit 1 no event record, bit 2 record but no
# wait on the communication stream, bit 4 always the same ring buffer
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp23; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so; cp gpurun_ab/lib_GX.so vpt_amd/libvpt_hip.so
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 400 --warmup 30 --check 0 --split-streams 1 --gather-root 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us")'
for h in 136 1080; do
$B --height $h --force-dist 0 2>/dev/null | python3 -c "$P" "H=$h plain" | tee -a "$out/ab.txt"
for x in 0 1 2 4 5; do
  VPT_GATHER_X=$x $B --height $h --force-dist 1 --gather native 2>"$out/err.txt" | python3 -c "$P" "H=$h gather X=$x" | tee -a "$out/ab.txt" || tail -5 "$out/err.txt"
done; done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
