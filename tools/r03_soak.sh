#!/bin/bash
# fuzz soak of the final library: the randomised differential test (tests/test_gpu_fuzz.py) over further seeds, every case on two
# streams (MCM: tile classes; marchers: HIT tiles only from their second frame on) and on one
set -o pipefail
mkdir -p gpurun_out/r03
for sp in 2 1; do
  VPT_FUZZ_SEEDS=${1:-40:540} VPT_FUZZ_SPLIT=$sp VPT_FUZZ_LAZY=${2:--1} timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_dos.py -x -q -m gpu > gpurun_out/r03/soak_split$sp.log 2>&1; rc=$?
  tail -2 gpurun_out/r03/soak_split$sp.log
  [ $rc -ne 0 ] && { grep -E "^E |FAILED" gpurun_out/r03/soak_split$sp.log | head -20; exit $rc; }
done
exit 0
