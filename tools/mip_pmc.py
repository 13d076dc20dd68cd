#!/usr/bin/env python3
"""MIP 256^3 @ 1080p at a chosen camera distance (argv[1], default 2.0) for counter collection under rocprofv3 --pmc."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402

dist = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
W, H = 1920, 1080
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(256, noise=48.0), 'linear')
cam = default_camera(W / H)
cam.transform.localTranslation = [0, 0, dist]
r = vpt_amd.MIPRenderer(ctx, gvol, cam, None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
r.reset()
for _ in range(30):
    r.render()
ctx.synchronize()
print("samples per frame", r.sample_count() / 30)
