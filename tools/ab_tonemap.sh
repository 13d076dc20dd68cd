#!/bin/bash
# A/B of two builds of the library (gpurun_ab/lib_A.so, lib_B.so) on the tone-map table form at 2160p and 4320p
set -e
cp vpt_amd/libvpt_hip.so /tmp/lib_keep.so
for round in 1 2; do
  for v in A B; do
    cp gpurun_ab/lib_$v.so vpt_amd/libvpt_hip.so
    python3 - "$v" <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import vpt_amd
from vpt_amd import _native as N
ctx = vpt_amd.Context(0)
for (w, h) in ((1920, 1080), (3840, 2160)):
    img = (np.random.default_rng(1).uniform(0, 4, size=(h, w, 4)) ** 2).astype(np.float16)
    for kind in ('artistic', 'aces'):
        tm = vpt_amd.ToneMapperFactory(kind)(ctx, img, {'resolution': (w, h)})
        tm.set_option(N.TONEMAPPER_OPTION_TABLE, N.TONEMAPPER_TABLE_ALWAYS)
        for _ in range(20): tm.render()
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(200): tm.render()
        ctx.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(sys.argv[1], kind, '%dx%d' % (w, h), '%.1f us' % (dt * 1e6), '%.0f GB/s' % (12.0 * w * h / dt / 1e9))
        tm.destroy()
PY
  done
done
cp /tmp/lib_keep.so vpt_amd/libvpt_hip.so
