#!/bin/bash
# soak of the final library: the fuzz generator over fresh seeds, every case once on one stream and once on three
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_soak; mkdir -p "$out"
export TMPDIR=/tmp
for sp in 1 3; do
  VPT_FUZZ_SEEDS=4640:5640 VPT_FUZZ_SPLIT=$sp timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider > "$out/soak_split$sp.txt" 2>&1; rc=$?; echo "split $sp exit $rc"; tail -2 "$out/soak_split$sp.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
