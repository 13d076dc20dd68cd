#!/bin/bash
# round-2 batch 4: boundary atlas — parity, timing on/off x exact/fast, PMC
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp4; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_math.py -x -q -m gpu > "$out/parity.txt" 2>&1; tail -5 "$out/parity.txt"
echo "== timing"
for round in 1 2 3; do
  for cfg in "--boundary-atlas 1 --fast-math 0" "--boundary-atlas 0 --fast-math 0" "--boundary-atlas 1 --fast-math 1" "--boundary-atlas 0 --fast-math 1"; do
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done
done
echo "== other volumes (atlas on, exact / fast)"
for vol in 128 1024; do for fm in 0 1; do
  timeout -k 5 300 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --volume $vol --fast-math $fm 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('vol $vol fast $fm', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/ab.txt"
done; done
echo "== PMC"
for fm in 0 1; do
cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 60 --warmup 10 --fast-math $fm"
for group in "VALUBusy" "TA_BUSY_avr TA_TA_BUSY_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    name=fm${fm}_$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 200 rocprofv3 --pmc $group -d "$out/pmc_$name" -o pmc --output-format csv -- $cmd > "$out/pmc_$name.log" 2>&1 && echo "pmc '$group' ok" || { echo "pmc '$group' FAILED"; tail -3 "$out/pmc_$name.log"; }
done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "r02_exp4")
for fm in (0, 1):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(out, "pmc_fm%d_*" % fm, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_mcm_integrate" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print("fast=%d %-30s mean/launch %.6g  (%d launches)" % (fm, k, s / n, n))
PY
