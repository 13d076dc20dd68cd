#!/bin/bash
# Round-4 evidence: rocprofv3 kernel statistics (+ PMC passes) of the shipped library; summaries are copied to gpurun_out/r04/profiles/ under
# the names profiles/ keeps them by.  Every *_kernel_stats.csv carries the traced run's own ms_per_step in its header.
#   tools/r04_profiles.sh 1 | 2 | 3 | 4     (four calls on the GPU pool, each within one box's time limit)
set -o pipefail
P=gpurun_out/r04/profiles; mkdir -p $P
run() {   # name, bench.py arguments
  name=$1; shift
  tools/pmc.sh r04 $name "$@" 2>&1 | grep -v amdgpu.ids
  [ -f gpurun_out/r04/pmc_$name/summary.json ] && cp gpurun_out/r04/pmc_$name/summary.json $P/r04_${name}_pmc.json
  cp gpurun_out/r04/pmc_$name/kernel_stats.csv $P/r04_${name}_kernel_stats.csv
}
part=${1:-1}
if [ $part = 1 ]; then
# the default line (only the arithmetic is set): HIT | MISS kernels overlapping on two streams (the PMC passes serialise the dispatches)
run mcm512_fast_default --fast-math 1
# the same two kernels ONE AFTER THE OTHER on one stream: each kernel alone on the chip, traced (no counters needed: kernel-trace only)
PMC=0 run mcm512_fast_each_kernel_alone --fast-math 1 --split-streams 1 --tile-classes 2
PMC=0 run mcm512_bit_exact_each_kernel_alone --fast-math 0 --split-streams 1 --tile-classes 2
run mcm512_bit_exact_default --fast-math 0
fi
if [ $part = 2 ]; then
run mcm1024_fast_default --fast-math 1 --volume 1024
run mcm1024_fast_bricks --fast-math 1 --volume 1024 --column-records 0
PMC=0 run mcm1024_fast_each_kernel_alone --fast-math 1 --volume 1024 --split-streams 1 --tile-classes 2
fi
if [ $part = 3 ]; then
PMC=0 run eam256_default_three_streams --renderer eam --volume 256
run eam256_one_stream --renderer eam --volume 256 --split-streams 1
PMC=0 run mcs512_default --renderer mcs
fi
if [ $part = 4 ]; then
# the other ray marchers on the library's defaults (kernel traces only)
PMC=0 run mip512_default --renderer mip
PMC=0 run iso512_default --renderer iso
PMC=0 run depth512_default --renderer depth
PMC=0 run lao512_default --renderer lao --steps 20 --warmup 3
fi
ls $P
