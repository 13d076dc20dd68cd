#!/bin/bash
# Round-2 evidence for the final library: rocprofv3 kernel statistics + PMC passes of the MCM pass in its forms, the 1/8-shard
# probe, the steps sweep and the default bench line.  Summaries land in gpurun_out/r02_final/summary/ (copied into profiles/).
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_final; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
base="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 100 --warmup 10 --warmup-seconds 0 --repeats 1"
declare -A CFG
CFG[mcm512_fast_three_streams]="--fast-math 1 --split-streams 3"
CFG[mcm512_fast_one_stream]="--fast-math 1 --split-streams 1"
CFG[mcm512_bit_exact_three_streams]="--fast-math 0 --split-streams 3"
CFG[mcm512_bit_exact_one_stream]="--fast-math 0 --split-streams 1"
CFG[mcm512_bit_exact_no_atlas]="--fast-math 0 --split-streams 1 --boundary-atlas 0"
for name in mcm512_fast_three_streams mcm512_fast_one_stream mcm512_bit_exact_three_streams mcm512_bit_exact_one_stream mcm512_bit_exact_no_atlas; do
  cmd="$base ${CFG[$name]}"
  d="$out/$name"; mkdir -p "$d"; echo "$cmd" > "$d/command.txt"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$d/kt" -o kt --output-format csv -- $cmd > "$d/kt.log" 2>&1 && echo "$name kernel-trace ok" || { echo "$name kernel-trace FAILED"; tail -3 "$d/kt.log"; }
  for group in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    g=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --pmc $group -d "$d/pmc_$g" -o pmc --output-format csv -- $cmd > "$d/pmc_$g.log" 2>&1 || { echo "$name pmc '$group' FAILED"; tail -2 "$d/pmc_$g.log"; }
  done
  echo "$name pmc done"
done
python3 tools/summarise_r02.py "$out"
echo "== shard8 probe"; timeout -k 5 400 python3 tools/shard8_probe.py "$out/summary/r02_shard8.json" > "$out/shard8.log" 2>&1; tail -12 "$out/shard8.log"
echo "== steps sweep"; timeout -k 5 300 python3 tools/mcm_steps_sweep.py 512 > "$out/summary/r02_mcm_steps_sweep_bit_exact.json" 2> "$out/steps_sweep.err"; tail -4 "$out/summary/r02_mcm_steps_sweep_bit_exact.json"
echo "== default bench line"; timeout -k 5 700 python3 bench.py > "$out/summary/r02_bench_default.json" 2> "$out/bench_default.err"; tail -c 1500 "$out/summary/r02_bench_default.json"; tail -2 "$out/bench_default.err"
