#!/bin/bash
# A/B of two builds of the library in ONE box: gpurun_ab/lib_A.so vs lib_B.so, alternating
set -e
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in A B; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    python3 tools/dos_rate.py --volume 256 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['us_per_slice'],2))"
  done
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
