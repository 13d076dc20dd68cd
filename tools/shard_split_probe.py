#!/usr/bin/env python3
"""Rank 3 of 8's share of the headline frame on one GPU, as 1, 2 and 3 tile-row ranges on as many HIP streams
(VPT_OPTION_SPLIT_STREAMS): how much of a shard's 21 us per frame is launch gap and tail?"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd import _native as N                                   # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402


def timed(ctx, step, frames=400, reps=5):
    for _ in range(50):
        step()
    ctx.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / frames
        best = dt if best is None else min(best, dt)
    return best * 1e6


def main():
    W, H = 1920, 1080
    ctx = vpt_amd.Context(0)
    gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
    out = {}
    for world in (8, 4, 2):
        for fast in (0, 1):
            for split in (1, 2, 3):
                sh = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng(), 'shard': (world // 2 - 1, world, 8)})
                sh.set_option(N.OPTION_FAST_MATH, fast); sh.set_option(N.OPTION_SPLIT_STREAMS, split); sh.reset()
                out["world%d_%s_split%d" % (world, "fast" if fast else "exact", split)] = timed(ctx, sh.render)
                sh.destroy()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
