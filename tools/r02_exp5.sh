#!/bin/bash
# round-2 batch 5: whole GPU suite on the current library; strided tile-row order on/off; two-stream probe
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp5; mkdir -p "$out"
export TMPDIR=/tmp
echo "== timing"
for round in 1 2 3; do
  for cfg in "--row-order 1 --fast-math 0" "--row-order 0 --fast-math 0" "--row-order 1 --fast-math 1" "--row-order 0 --fast-math 1"; do
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done
done
for r in mip eam mcs; do for ro in 1 0; do
  timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --renderer $r --volume 256 --row-order $ro 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$r 256 row-order $ro', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/ab.txt"
done; done
echo "== two-stream probe"; timeout -k 5 300 python3 tools/two_stream_probe.py 512 1 2>&1 | tee "$out/two_stream_fast.txt" | tail -5
echo "== GPU suite"
timeout -k 5 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_configs.py > "$out/suite.txt" 2>&1; tail -8 "$out/suite.txt"
