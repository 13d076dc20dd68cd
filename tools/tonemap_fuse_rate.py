#!/usr/bin/env python3
"""RenderingContext.render() = renderer.render(); toneMapper.render() at the headline size (MCM 512^3, 1920x1080, fast-math, tile classes
on two streams; default Artistic tone mapper): us per displayed frame with VPT_TONEMAPPER_OPTION_FUSE off / on, and the renderer alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                  # noqa: E402
import vpt_amd                                                      # noqa: E402
from vpt_amd import _native as N                                    # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera           # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume         # noqa: E402

W, H = 1920, 1080
cache = "/tmp/vpt_vol_512.npy"
vol = np.load(cache) if os.path.exists(cache) else sphere_volume(512, noise=48.0)
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
for kind in (sys.argv[1:] or ["artistic", "reinhard"]):
    for fuse in (None, 0, 1):
        r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
        r.set_option(N.OPTION_FAST_MATH, 1); r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.reset()
        tm = None
        if fuse is not None:
            tm = vpt_amd.ToneMapperFactory(kind)(ctx, r, {'resolution': (W, H)})
            tm.set_option(N.TONEMAPPER_OPTION_FUSE, fuse)

        def frame():
            r.render()
            if tm is not None:
                tm.render()
        for _ in range(300):
            frame()
        ctx.synchronize()
        blocks = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(300):
                frame()
            ctx.synchronize()
            blocks.append((time.perf_counter() - t0) / 300 * 1e6)
        blocks.sort()
        print("%-10s %-28s %7.2f us per displayed frame" % (kind, "renderer alone" if fuse is None else ("tone mapper, fuse %d" % fuse), blocks[2]), flush=True)
        if tm is not None:
            tm.destroy()
        r.destroy()
gvol.destroy(); ctx.destroy()
