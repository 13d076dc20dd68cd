#!/bin/bash
# K-way split streams; the tail of a launch (time against the number of tile rows); early reset at 5 waves with K streams
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp14; mkdir -p "$out"
export TMPDIR=/tmp
B="timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us")'
echo "== parity"; timeout -k 5 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split" > "$out/parity.txt" 2>&1; tail -3 "$out/parity.txt"
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2; do
  for v in BASE ER5; do for cfg in "--fast-math 0 --split-streams 1" "--fast-math 0 --split-streams 2" "--fast-math 0 --split-streams 3" "--fast-math 0 --split-streams 4" "--fast-math 1 --split-streams 1" "--fast-math 1 --split-streams 2" "--fast-math 1 --split-streams 3" "--fast-math 1 --split-streams 4"; do
    [ $v = ER5 ] && [[ "$cfg" == *"math 0"* ]] && continue
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    $B $cfg 2>/dev/null | python3 -c "$P" "$v $cfg" | tee -a "$out/ab.txt"
  done; done
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
echo "== tail: one stream, fast, tile rows"
for h in 944 1072 1080 1184 1200 1424 1440 2160; do
  $B --fast-math 1 --split-streams 1 --height $h 2>/dev/null | python3 -c "$P" "H $h rows $(( (h+15)/16 ))" | tee -a "$out/tail.txt"
done
