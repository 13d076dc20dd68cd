#!/bin/bash
# VPT_PLAY_FRAMES as 1 / 3 tile-row ranges on as many streams
set -o pipefail
root=$(pwd); out=$root/gpurun_out/r02_exp37; mkdir -p "$out"
export TMPDIR=/tmp
echo "== parity"; timeout -k 5 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "play or frames or split" > "$out/parity.txt" 2>&1; echo "exit $?"; tail -3 "$out/parity.txt"
timeout -k 5 300 python3 - > "$out/frames.txt" 2>&1 <<'PY'
import sys, time; sys.path.insert(0, '.')
import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
W, H = 1920, 1080
for rep in range(2):
  for fm in (0, 1):
    for split in (1, 2, 3):
        for n in (8, 16):
            r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
            r.set_option(N.OPTION_FAST_MATH, fm); r.set_option(N.OPTION_SPLIT_STREAMS, split); r.reset()
            for _ in range(6): r.play(n, frames=True)
            ctx.synchronize()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(192 // n): r.play(n, frames=True)
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / 192
                best = dt if best is None else min(best, dt)
            print("fast" if fm else "exact", "split", split, n, "per launch:", round(best * 1e6, 2), "us per frame", "frac %.3f" % (24 * 16588800 / best / 8e12))
            r.destroy()
PY
cat "$out/frames.txt" | grep -v amdgpu.ids
