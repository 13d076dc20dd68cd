#!/bin/bash
# bench.py driving VPT_PLAY_FRAMES (--frames-per-launch 16 --fused-passes 2): the line, with its oracle frame check
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp38; mkdir -p "$out"
export TMPDIR=/tmp
for fm in 1 0; do
timeout -k 5 400 python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --frames-per-launch 16 --fused-passes 2 --fast-math $fm > "$out/line_fm$fm.json" 2>"$out/err.txt"; echo "exit $?"
python3 -c "
import json; d=json.load(open('$out/line_fm$fm.json')); print(d['ms_per_step'], d['roofline']['frac'], d['frame_check'], d.get('frame_check_kind'), d['config']['frames_per_launch'], d['config']['every_frame_written'])"
done
