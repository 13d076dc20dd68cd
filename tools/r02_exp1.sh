#!/bin/bash
# round-2 experiment batch 1 (run through gpurun from the repo root): instruction-issue and line-gather microbenchmarks,
# PMC passes on the headline kernel, A/B of the experiment builds in gpurun_ab/
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp1; mkdir -p "$out"
export TMPDIR=/tmp
echo "== counters"; (rocprofv3 -L > "$out/counters.txt" 2>&1 || rocprofv3 --list-avail > "$out/counters.txt" 2>&1); wc -l "$out/counters.txt"
echo "== valu_rates"; timeout -k 5 120 ./tools/valu_rates > "$out/valu_rates.txt" 2>&1; tail -3 "$out/valu_rates.txt"
echo "== line_gather"; timeout -k 5 120 ./tools/line_gather > "$out/line_gather.txt" 2>&1; tail -3 "$out/line_gather.txt"
echo "== A/B"
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2 3; do
  for v in R1 X0 X2 X4 X8 X6 X14; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 200 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step']*1e3,2), 'us', 'kernel', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$out/ab.txt"
  done
done
echo "== parity of the X14 build (merged draws + aligned taps)"
cp gpurun_ab/lib_X14.so vpt_amd/libvpt_hip.so
timeout -k 5 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$out/parity_X14.txt" 2>&1; tail -2 "$out/parity_X14.txt"
cp gpurun_ab/lib_X0.so vpt_amd/libvpt_hip.so
echo "== PMC on X0"
cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 60 --warmup 10"
for group in "VALUBusy MemUnitBusy" "MemUnitStalled L2CacheHit" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    name=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 200 rocprofv3 --pmc $group -d "$out/pmc_$name" -o pmc --output-format csv -- $cmd > "$out/pmc_$name.log" 2>&1 && echo "pmc '$group' ok" || { echo "pmc '$group' FAILED"; tail -3 "$out/pmc_$name.log"; }
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.getcwd(), "gpurun_out", "r02_exp1")
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
