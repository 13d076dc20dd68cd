#!/usr/bin/env python3
"""MIP frame time against steps for the default camera (cube on ~20 % of the image) and a close camera (whole image)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402

W, H = 1920, 1080
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(256, noise=48.0), 'linear')
for dist in (2.0, 1.0):
    cam = default_camera(W / H)
    cam.transform.localTranslation = [0, 0, dist]
    r = vpt_amd.MIPRenderer(ctx, gvol, cam, None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    for steps in (1, 16, 32, 64, 128, 256):
        r.steps = steps
        r.reset()
        for _ in range(20):
            r.render()
        ctx.synchronize(); r.clear_sample_count()
        t0 = time.perf_counter()
        for _ in range(200):
            r.render()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print("z=%.1f steps %3d  %.4f ms  %.3g samples/s" % (dist, steps, dt * 1e3, r.sample_count() / 200 / dt), flush=True)
    r.destroy()
