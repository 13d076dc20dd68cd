#!/bin/bash
# round-2 experiment batch 2: instruction costs (new kernels), the fast-math variant (tests + timing), extraction fix, two-stream probe
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp2; mkdir -p "$out"
export TMPDIR=/tmp
echo "== valu_rates"; timeout -k 5 200 ./tools/valu_rates > "$out/valu_rates.txt" 2>&1; grep -c cycles "$out/valu_rates.txt"
echo "== fast-math tests"; timeout -k 5 400 python3 -m pytest tests/test_gpu_fast_math.py -x -q -m gpu > "$out/fast_tests.txt" 2>&1; tail -15 "$out/fast_tests.txt"
echo "== parity (extraction fix)"; timeout -k 5 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$out/parity.txt" 2>&1; tail -3 "$out/parity.txt"
echo "== A/B"
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in R1 X0 NEW X32; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 200 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done
  cp gpurun_ab/lib_NEW.so vpt_amd/libvpt_hip.so
  timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 200 --warmup 30 --fast-math 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW-fast', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
echo "== two-stream probe (exact)"; timeout -k 5 300 python3 tools/two_stream_probe.py 512 0 2>&1 | tee "$out/two_stream_exact.txt" | tail -5
echo "== two-stream probe (fast)"; timeout -k 5 300 python3 tools/two_stream_probe.py 512 1 2>&1 | tee "$out/two_stream_fast.txt" | tail -5
echo "== PMC fast"
cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 60 --warmup 10 --fast-math 1"
for group in "VALUBusy MemUnitBusy" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    name=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 200 rocprofv3 --pmc $group -d "$out/pmc_$name" -o pmc --output-format csv -- $cmd > "$out/pmc_$name.log" 2>&1 && echo "pmc '$group' ok" || { echo "pmc '$group' FAILED"; tail -3 "$out/pmc_$name.log"; }
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "r02_exp2")
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_mcm_integrate" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open(os.path.join(out, "pmc_summary.txt"), "w") as f:
    for k, (s, n) in sorted(acc.items()):
        line = "%-36s mean/launch %.6g  (%d launches)" % (k, s / n, n)
        print(line); f.write(line + "\n")
PY
