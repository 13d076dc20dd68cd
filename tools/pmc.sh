#!/bin/bash
# rocprofv3 kernel statistics + PMC passes (one counter group per run, never mixed with a trace) of one bench.py configuration:
#   tools/pmc.sh <round> <name> [bench.py arguments]        -> gpurun_out/<round>/pmc_<name>/summary.json, kernel_stats.csv
set -o pipefail
round=$1; name=$2; shift; shift
out=gpurun_out/$round/pmc_$name; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --kernels-alone 0 --check 0 --steps 100 --warmup 10 --warmup-seconds 0 --repeats 1 $*"
echo "$cmd" > $out/command.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- $cmd > $out/kt.log 2>&1 || { echo "kernel-trace FAILED"; tail -3 $out/kt.log; exit 1; }
if [ "${PMC:-1}" = 1 ]; then
for group in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  g=$(echo "$group" | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $group -d $out/pmc_$g -o pmc --output-format csv -- $cmd > $out/pmc_$g.log 2>&1 || { echo "pmc '$group' FAILED"; tail -2 $out/pmc_$g.log; }
done
fi
python3 tools/summarise_pmc.py $out
rm -rf $out/kt $out/pmc_*/
