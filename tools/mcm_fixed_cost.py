#!/usr/bin/env python3
"""steps = 0 and steps = 8 MCM frame time at the headline size (used with experimental builds of the library)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402

W, H = 1920, 1080
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(128, noise=48.0), 'linear')
r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
for steps in (0, 8):
    r.steps = steps
    r.reset()
    for _ in range(30):
        r.render()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        r.render()
    ctx.synchronize()
    print(sys.argv[1] if len(sys.argv) > 1 else "", "steps", steps, "%.4f ms" % ((time.perf_counter() - t0) / 400 * 1e3), flush=True)
