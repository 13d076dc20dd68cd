#!/bin/bash
# round-2 evidence for profiles/: rocprofv3 kernel-trace statistics and PMC passes (one --pmc run per counter group, never mixed
# with traces) for the headline workload in its forms, the ray marchers (MIP / EAM 256^3) and the 1/8-shard probe.
# Run through gpurun from the repo root; tools/summarise_r02.py condenses the raw output into gpurun_out/r02_prof/summary/.
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_prof; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
echo "== waves per SIMD A/B"
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2; do for v in NEW W6 W5; do for fm in 0 1; do
  cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
  timeout -k 5 200 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 200 --warmup 30 --fast-math $fm --split-streams 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v fast $fm', round(d['ms_per_step']*1e3,2), 'us')" | tee -a "$out/waves_ab.txt"
done; done; done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
base="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 100 --warmup 10 --warmup-seconds 0 --repeats 1"
declare -A CFG
CFG[mcm512_fast_two_streams]="--fast-math 1 --split-streams 2"
CFG[mcm512_fast_one_stream]="--fast-math 1 --split-streams 1"
CFG[mcm512_bit_exact_one_stream]="--fast-math 0 --split-streams 1"
CFG[mcm512_bit_exact_no_atlas]="--fast-math 0 --split-streams 1 --boundary-atlas 0"
CFG[mip256]="--renderer mip --volume 256"
CFG[eam256]="--renderer eam --volume 256"
for name in mcm512_fast_two_streams mcm512_fast_one_stream mcm512_bit_exact_one_stream mcm512_bit_exact_no_atlas mip256 eam256; do
  cmd="$base ${CFG[$name]}"
  d="$out/$name"; mkdir -p "$d"; echo "$cmd" > "$d/command.txt"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$d/kt" -o kt --output-format csv -- $cmd > "$d/kt.log" 2>&1 && echo "$name kernel-trace ok" || { echo "$name kernel-trace FAILED"; tail -3 "$d/kt.log"; }
  for group in "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    g=$(echo "$group" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --pmc $group -d "$d/pmc_$g" -o pmc --output-format csv -- $cmd > "$d/pmc_$g.log" 2>&1 || { echo "$name pmc '$group' FAILED"; tail -2 "$d/pmc_$g.log"; }
  done
  echo "$name pmc done"
done
echo "== shard8 probe"; timeout -k 5 400 python3 tools/shard8_probe.py "$out/shard8.json" > "$out/shard8.log" 2>&1; tail -12 "$out/shard8.log"
echo "== steps sweep"; timeout -k 5 300 python3 tools/mcm_steps_sweep.py 512 > "$out/steps_sweep_512.json" 2> "$out/steps_sweep.err"; tail -3 "$out/steps_sweep_512.json"
python3 tools/summarise_r02.py "$out"
