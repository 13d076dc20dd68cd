#!/bin/bash
# rocprofv3 kernel statistics + PMC passes of the MIP and EAM marchers at 256^3, 1080p (one stream), summaries into summary/
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_marchers; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
base="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 100 --warmup 10 --warmup-seconds 0 --repeats 1 --split-streams 1"
declare -A CFG
CFG[mip256]="--renderer mip --volume 256"
CFG[eam256]="--renderer eam --volume 256"
for name in mip256 eam256; do
  cmd="$base ${CFG[$name]}"
  d="$out/$name"; mkdir -p "$d"; echo "$cmd" > "$d/command.txt"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$d/kt" -o kt --output-format csv -- $cmd > "$d/kt.log" 2>&1 && echo "$name kernel-trace ok" || { echo "$name kernel-trace FAILED"; tail -3 "$d/kt.log"; }
  for group in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    g=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --pmc $group -d "$d/pmc_$g" -o pmc --output-format csv -- $cmd > "$d/pmc_$g.log" 2>&1 || { echo "$name pmc '$group' FAILED"; tail -2 "$d/pmc_$g.log"; }
  done
done
python3 tools/summarise_r02.py "$out"
