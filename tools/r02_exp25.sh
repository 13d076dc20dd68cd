#!/bin/bash
# the latency-bound regime (a 1/8, 1/4, 1/2 shard of the headline frame on one GPU): path end of out-of-cube lanes while the
# sample's loads fly (experiment build, 96 / 128 VGPRs) and plain 4-wave builds against the shipped library
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp25; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for v in BASE ER5 ER4 W4; do
  cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
  echo "== $v"; timeout -k 5 400 python3 tools/shard_split_probe.py 2>/dev/null | grep fast | tr -d '\n'; echo
done | tee "$out/shards.txt"
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
