#!/usr/bin/env python3
"""Samples/s of the ray-marching renderers with the cube covering ~25 % of the image (default camera) and the whole
image (camera pulled in): how much of their frame time is image-space load imbalance."""
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
for kind in ("mip", "eam", "iso", "depth"):
    for dist in (2.0, 1.0, 0.55):
        cam = default_camera(W / H)
        cam.transform.localTranslation = [0, 0, dist]
        r = vpt_amd.RendererFactory(kind)(ctx, gvol, cam, None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
        r.reset()
        for _ in range(20):
            r.render()
        ctx.synchronize(); r.clear_sample_count()
        t0 = time.perf_counter()
        for _ in range(200):
            r.render()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 200
        ns = r.sample_count() / 200
        print("%-5s camera z=%.2f  %.4f ms/frame  %.3g samples/frame  %.3g samples/s" % (kind, dist, dt * 1e3, ns, ns / dt), flush=True)
        r.destroy()
