#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_prof2; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
base="python3 bench.py --cpu-baseline 0 --stream-probe 0 --other-configs 0 --check 0 --steps 100 --warmup 10 --warmup-seconds 0 --repeats 1"
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
echo "== MCS fused probe"
python3 - <<'PY'
import time, sys, os
sys.path.insert(0, os.getcwd())
import vpt_amd
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
r = vpt_amd.MCSRenderer(ctx, gvol, default_camera(1920 / 1080), None, {'resolution': (1920, 1080), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
r.reset()
for n in (1, 4, 16):
    for _ in range(3): r.play(n, fused=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    u, v = None, None
    for _ in range(10): u, v = r._collect_frames(n)
    t_py = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(20): r.play(n, fused=True)
    t_enq = (time.perf_counter() - t0) / 20
    ctx.synchronize()
    t_all = (time.perf_counter() - t0) / 20
    print("MCS fused n=%d: collect %.3f ms, enqueue %.3f ms, total %.3f ms per call (%.4f per pass)" % (n, t_py * 1e3, t_enq * 1e3, t_all * 1e3, t_all * 1e3 / n))
PY
echo "== JS gpu test"; timeout -k 5 300 python3 -m pytest tests/test_js_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== default bench line"; timeout -k 5 600 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -c 2500 "$out/bench_default.json"; tail -2 "$out/bench_default.err"
