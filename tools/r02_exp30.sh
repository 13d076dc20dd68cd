#!/bin/bash
# split streams on the other sampling renderers: parity, then MIP / EAM / MCS / ISO at 256^3 and 512^3, 1 vs 2 vs 3 streams
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp30; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
B="timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"]*1e3,2), "us", "%.3g" % d["value"])'
for cfg in "--renderer eam --volume 256" "--renderer mip --volume 256" "--renderer mcs --volume 512" "--renderer iso --volume 256" "--renderer eam --volume 512"; do for sp in 1 2 3; do
  $B $cfg --split-streams $sp 2>/dev/null | python3 -c "$P" "$cfg split=$sp" | tee -a "$out/ab.txt"
done; done
