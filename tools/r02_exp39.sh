#!/bin/bash
# MCM as one-wave workgroups (a wave slot is free as soon as ITS wave ends, not when the slowest of four does)?
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp39; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so; cp gpurun_ab/lib_WB.so vpt_amd/libvpt_hip.so
B="timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", d.get("frame_check"))'
for rep in 1 2; do for wb in 0 1; do for cfg in "--fast-math 1 --split-streams 1" "--fast-math 1 --split-streams 3" "--fast-math 0 --split-streams 3" "--fast-math 1 --split-streams 3 --volume 256"; do
  VPT_MCM_WAVE_BLOCKS=$wb $B $cfg 2>/dev/null | python3 -c "$P" "wave_blocks=$wb $cfg" | tee -a "$out/ab.txt"
done; done; done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
