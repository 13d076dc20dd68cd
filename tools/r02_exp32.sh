#!/bin/bash
# more hardware queues (GPU_MAX_HW_QUEUES) so that 4..8 ranges each get a queue of their own?
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp32; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so; cp gpurun_ab/lib_S8.so vpt_amd/libvpt_hip.so
B="timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us")'
for q in 4 8; do for k in 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q $B --split-streams $k 2>/dev/null | python3 -c "$P" "queues=$q fast K=$k" | tee -a "$out/ab.txt"
done; done
for q in 8; do for k in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q $B --split-streams $k --fast-math 0 2>/dev/null | python3 -c "$P" "queues=$q exact K=$k" | tee -a "$out/ab.txt"
  GPU_MAX_HW_QUEUES=$q $B --split-streams $k --renderer eam --volume 256 2>/dev/null | python3 -c "$P" "queues=$q eam K=$k" | tee -a "$out/ab.txt"
done; done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
