#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp8; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity skipped"
echo "== timing"
for round in 1 2 3; do
  for cfg in "--fast-math 0 --split-streams 1" "--fast-math 0 --split-streams 2" "--fast-math 1 --split-streams 1" "--fast-math 1 --split-streams 2"; do
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done
done
echo "== full line with check, split 2 fast"
timeout -k 5 400 python3 bench.py --fast-math 1 --split-streams 2 --other-configs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['frame_check'], d.get('frame_check_kind'), d['roofline']['frac'])"
echo "== other volumes split 2"
for vol in 128 1024; do for fm in 0 1; do
  timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --volume $vol --fast-math $fm --split-streams 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('split2 vol $vol fast $fm', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/ab.txt"
done; done
