#!/bin/bash
# round 4, first GPU call: parity of the column records and of the marchers' destination tracking, then the records A/B (H and C4)
set -o pipefail
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_records.py tests/test_gpu_marcher_classes.py tests/test_gpu_tile_classes.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -15 > $out/tests_a.log; rc=${PIPESTATUS[0]}
cat $out/tests_a.log
[ $rc -eq 0 ] || exit 1
{
for rec in 0 1 0 1; do
  timeout -k 10 300 python3 tools/ab_mcm.py --split 2 --records $rec --fast 1 --kernel-times 1 --digest 1 --tag "H fast" --blocks 3 || exit 1
done
for rec in 0 1; do
  timeout -k 10 300 python3 tools/ab_mcm.py --split 2 --records $rec --fast 0 --kernel-times 1 --digest 1 --tag "H bit-exact" --blocks 3 || exit 1
  timeout -k 10 300 python3 tools/ab_mcm.py --split 1 --records $rec --fast 1 --kernel-times 1 --tag "H fast one stream (kernels alone)" --blocks 3 || exit 1
  timeout -k 10 300 python3 tools/ab_mcm.py --split 1 --classes 0 --records $rec --fast 1 --tag "H fast general kernel" --blocks 3 || exit 1
  timeout -k 10 300 python3 tools/ab_mcm.py --split 2 --records $rec --fast 1 --camera-z 0.9 --kernel-times 1 --tag "H fast all-HIT camera" --blocks 3 || exit 1
  timeout -k 10 300 python3 tools/ab_mcm.py --split 2 --records $rec --fast 1 --extinction 50 --tf ramp --kernel-times 1 --tag "H fast ext 50 ramp" --blocks 3 || exit 1
  timeout -k 10 300 python3 tools/ab_mcm.py --split 2 --records $rec --fast 1 --shard 3,8,8 --kernel-times 1 --tag "H/8 fast" --blocks 3 || exit 1
done
for rec in 0 1 0 1; do
  timeout -k 10 400 python3 tools/ab_mcm.py --volume 1024 --split 2 --records $rec --fast 1 --kernel-times 1 --digest 1 --tag "C4 fast" --blocks 3 || exit 1
done
for rec in 0 1; do
  timeout -k 10 400 python3 tools/ab_mcm.py --volume 1024 --split 2 --records $rec --fast 0 --kernel-times 1 --tag "C4 bit-exact" --blocks 3 || exit 1
  timeout -k 10 400 python3 tools/ab_mcm.py --volume 1024 --split 1 --records $rec --fast 1 --kernel-times 1 --tag "C4 fast one stream (kernels alone)" --blocks 3 || exit 1
done
} 2>&1 | grep -v amdgpu | tee $out/records_ab.log
