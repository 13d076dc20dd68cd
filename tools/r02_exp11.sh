#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp11; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in NEW OV GT OVGT; do for cfg in "--fast-math 0 --split-streams 1" "--fast-math 1 --split-streams 1" "--fast-math 1 --split-streams 2"; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $cfg', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/ab.txt"
  done; done
done
cp gpurun_ab/lib_OVGT.so vpt_amd/libvpt_hip.so
echo "== OVGT parity"; timeout -k 5 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_math.py -x -q -m gpu > "$out/parity.txt" 2>&1; tail -2 "$out/parity.txt"
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
