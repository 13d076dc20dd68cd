#!/bin/bash
# Round-3 evidence: rocprofv3 kernel statistics + PMC passes (tools/r03_pmc.sh) of the headline workload in its forms and of the
# ray marchers; summaries are copied to gpurun_out/r03/profiles/ under the names profiles/ keeps them by.
set -o pipefail
P=gpurun_out/r03/profiles; mkdir -p $P
run() {   # name, bench.py arguments
  name=$1; shift
  tools/r03_pmc.sh $name "$@" 2>&1 | grep -v amdgpu.ids
  cp gpurun_out/r03/pmc_$name/summary.json $P/r03_${name}_pmc.json
  cp gpurun_out/r03/pmc_$name/kernel_stats.csv $P/r03_${name}_kernel_stats.csv
}
# (tile classes run where two streams are allowed; counter collection serialises the dispatches, so the PMC passes of the two-stream
# form see each kernel alone on the chip — the kernel-trace pass of the same directory shows them overlapping)
# (two calls on the GPU pool: `r03_profiles.sh 1` and `r03_profiles.sh 2`, each within one box's time limit)
part=${1:-all}
if [ $part = all ] || [ $part = 1 ]; then
run mcm512_fast_classes --fast-math 1 --split-streams 2
run mcm512_bit_exact_classes --fast-math 0 --split-streams 2
run mcm512_fast_general_kernel_one_stream --fast-math 1 --split-streams 1 --tile-classes 0
fi
[ $part = 1 ] && { ls $P; exit 0; }
run eam256_classes_one_stream --renderer eam --volume 256 --split-streams 1
run mip256_classes_one_stream --renderer mip --volume 256 --split-streams 1
# a kernel trace of the marchers' three-stream form
export TMPDIR=/tmp
for cfg in "eam256_classes_three_streams:--renderer eam --volume 256 --split-streams 3"; do
  name=${cfg%%:*}; args=${cfg#*:}
  d=gpurun_out/r03/kt_$name; rm -rf $d; mkdir -p $d
  cmd="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 100 --warmup 10 --warmup-seconds 0 --repeats 1 $args"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d/kt -o kt --output-format csv -- $cmd > $d/kt.log 2>&1 || { echo "$name kernel-trace FAILED"; tail -3 $d/kt.log; continue; }
  echo "# $cmd" > $P/r03_${name}_kernel_stats.csv
  python3 tools/kernel_stats.py $d/kt | head -8 >> $P/r03_${name}_kernel_stats.csv
  rm -rf $d/kt
done
ls $P
