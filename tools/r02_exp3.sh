#!/bin/bash
# round-2 experiment batch 3: texture-path costs, speculative resetPhoton (X128), cheap out-of-cube sample hack (X64, timing only),
# the new configuration tests
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp3; mkdir -p "$out"
export TMPDIR=/tmp
echo "== gather_rates"; timeout -k 5 200 ./tools/gather_rates > "$out/gather_rates.txt" 2>&1; tail -3 "$out/gather_rates.txt"
echo "== A/B"
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in R1 NEW X64 X128 X192; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done
done
echo "== parity X128"
cp gpurun_ab/lib_X128.so vpt_amd/libvpt_hip.so
timeout -k 5 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mcm or gather" > "$out/parity_X128.txt" 2>&1; tail -3 "$out/parity_X128.txt"
echo "== PMC X128"
cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 60 --warmup 10"
for group in "VALUBusy" "TA_BUSY_avr TA_TA_BUSY_sum"; do
    name=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 200 rocprofv3 --pmc $group -d "$out/pmc_$name" -o pmc --output-format csv -- $cmd > "$out/pmc_$name.log" 2>&1 && echo "pmc '$group' ok" || { echo "pmc '$group' FAILED"; tail -3 "$out/pmc_$name.log"; }
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
echo "== full bench line (default lib)"
timeout -k 5 500 python3 bench.py > "$out/bench_full.json" 2> "$out/bench_full.err"; tail -c 3000 "$out/bench_full.json"; tail -3 "$out/bench_full.err"
echo "== config tests"
timeout -k 5 900 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu > "$out/configs.txt" 2>&1; tail -15 "$out/configs.txt"
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "r02_exp3")
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_mcm_integrate" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-36s mean/launch %.6g  (%d launches)" % (k, s / n, n))
PY
