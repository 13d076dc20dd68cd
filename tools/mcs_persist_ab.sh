#!/bin/bash
# MCS 512^3 @ 1080p: one thread per pixel (HIT tiles only, library defaults) against the persistent waves with ballot compaction, by extinction
mkdir -p gpurun_out/r04b; out=gpurun_out/r04b/mcs_persist.txt; : > $out
for e in 1 10 50 200 1000 4000; do
  for p in 0 1; do
    timeout -k 10 200 python3 tools/ab_mcm.py --renderer mcs --extinction $e --persistent $p --frames 100 --blocks 3 --tag "ext$e" 2>/dev/null >> $out || exit 1
  done
done
cat $out
