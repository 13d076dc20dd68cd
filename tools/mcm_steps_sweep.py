#!/usr/bin/env python3
"""MCM frame time against `steps` (events per pixel per pass) at the headline size: separates the per-pass fixed cost
(launch, LDS staging, photon-state stream, per-pixel set-up, tail) from the per-event cost."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402


def main():
    n, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 1920, 1080
    ctx = vpt_amd.Context(0)
    gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(n, noise=48.0), 'linear')
    out = {}
    for shard in (None, (3, 8, 8)):
        o = {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
        if shard:
            o['shard'] = shard
        r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, o)
        for steps in (0, 1, 2, 4, 8, 16):
            r.steps = steps
            r.reset()
            for _ in range(20):
                r.render()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                r.render()
            ctx.synchronize()
            out["%s_steps%d" % ("shard3of8" if shard else "full", steps)] = (time.perf_counter() - t0) / 300 * 1e3
        r.destroy()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
