#!/bin/bash
# slab fma form + folded log scale in the fast variant: tolerance tests, then timing at 1 and 3 streams
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp15; mkdir -p "$out"
export TMPDIR=/tmp
B="timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us")'
echo "== parity"; timeout -k 5 600 python3 -m pytest tests/test_gpu_fast_math.py tests/test_gpu_parity.py -x -q -m gpu > "$out/parity.txt" 2>&1; tail -3 "$out/parity.txt"
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in BASE NEW; do for cfg in "--fast-math 1 --split-streams 1" "--fast-math 1 --split-streams 3"; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    $B $cfg 2>/dev/null | python3 -c "$P" "$v $cfg" | tee -a "$out/ab.txt"
  done; done
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
