#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp7; mkdir -p "$out"
export TMPDIR=/tmp
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in NEW W8; do for fm in 0 1; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --fast-math $fm 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v fast $fm', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done; done
done
cp gpurun_ab/lib_W8.so vpt_amd/libvpt_hip.so
echo "== W8 parity"; timeout -k 5 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mcm" > "$out/parity_w8.txt" 2>&1; tail -2 "$out/parity_w8.txt"
echo "== W8 two-stream"; timeout -k 5 300 python3 tools/two_stream_probe.py 512 1 2>&1 | tail -5
echo "== W8 other volumes"
for vol in 128 1024; do for fm in 0 1; do
  timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --volume $vol --fast-math $fm 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('W8 vol $vol fast $fm', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/ab.txt"
done; done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
