#!/usr/bin/env python3
"""Frame time of the headline MCM workload against the phase-stagger quantum / pattern (VPT_OPTION_MCM_STAGGER)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd import _native as N                                   # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402


def main():
    n, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 1920, 1080
    ctx = vpt_amd.Context(0)
    gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(n, noise=48.0), 'linear')
    r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    out = {}
    for pattern in (0, 1):
        for us in (0, 2, 4, 6, 8, 10, 14, 20):
            r.set_option(N.OPTION_MCM_STAGGER, int(us * 100) | (pattern << 24))
            r.reset()
            for _ in range(30):
                r.render()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                r.render()
            ctx.synchronize()
            out["pattern%d_quantum%dus" % (pattern, us)] = round((time.perf_counter() - t0) / 300 * 1e3, 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
