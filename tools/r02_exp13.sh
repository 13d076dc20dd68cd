#!/bin/bash
# A/B: ending out-of-cube paths while the sample's loads fly (VPT_FAST_EARLY_RESET) at 7/6/5 waves per SIMD
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp13; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
cp gpurun_ab/lib_ER5.so vpt_amd/libvpt_hip.so
echo "== parity (ER5)"; timeout -k 5 600 python3 -m pytest tests/test_gpu_fast_math.py tests/test_gpu_parity.py -x -q -m gpu > "$out/parity.txt" 2>&1; tail -3 "$out/parity.txt"
for round in 1 2 3; do
  for v in S56 W5 ER7 ER6 ER5; do for cfg in "--fast-math 1 --split-streams 1" "--fast-math 1 --split-streams 2"; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $cfg', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/ab.txt"
  done; done
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
