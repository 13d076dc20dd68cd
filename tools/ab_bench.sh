#!/bin/bash
# A/B of two builds of the library in ONE box through bench.py: gpurun_ab/lib_A.so vs lib_B.so, alternating (the directory
# is git-ignored but travels with gpurun; copy the two builds of vpt_amd/libvpt_hip.so there by hand)
#   tools/ab_bench.sh --renderer mip --volume 256
set -e
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in A B; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    python3 bench.py --cpu-baseline 0 --stream-probe 0 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,2), 'us', '%.3g' % d['value'])"
  done
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
