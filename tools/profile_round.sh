#!/bin/bash
# Profiles the default bench.py workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>        e.g. r01b
# pass 1: rocprofv3 --kernel-trace --stats          -> per-kernel durations
# pass 2..n: one --pmc run per counter group         -> HBM traffic / VALU / occupancy counters (never mixed with traces)
# Raw output lands in gpurun_out/prof_<tag>/ (scratch); tools/summarise_profile.py condenses it into profiles/.
set -o pipefail
tag=${1:-round}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --steps 100 --warmup 10"
cd "$root" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- $cmd > "$out/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$out/kt.log"; exit 1; }
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32"; do
    name=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --pmc $group -d "$out/pmc_$name" -o pmc -- $cmd > "$out/pmc_$name.log" 2>&1 || { echo "pmc pass '$group' failed"; tail -5 "$out/pmc_$name.log"; exit 1; }
    echo "pmc pass '$group' done"
done
python3 tools/summarise_profile.py "$out" "$tag"
