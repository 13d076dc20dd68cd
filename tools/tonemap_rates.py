#!/usr/bin/env python3
"""Times the ten tone-map kernels on one GPU (wall time per render() over back-to-back launches, inputs in HBM).
Algorithmic bytes per pixel: 8 (RGBA16F read) + 4 (RGBA8 write) = 12."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd import _native as N                                   # noqa: E402

KINDS = ['artistic', 'range', 'reinhard', 'reinhard2', 'uncharted2', 'filmic', 'unreal', 'aces', 'lottes', 'uchimura']


def main():
    ctx = vpt_amd.Context(0)
    out = {}
    for (w, h) in ((1920, 1080), (3840, 2160), (7680, 4320)):
        img = (np.random.default_rng(1).uniform(0, 4, size=(h, w, 4)) ** 2).astype(np.float16)
        for kind in KINDS:
            tm = vpt_amd.ToneMapperFactory(kind)(ctx, img, {'resolution': (w, h)})
            rec = {}
            for name, mode in (("direct", N.TONEMAPPER_TABLE_NEVER), ("table", N.TONEMAPPER_TABLE_ALWAYS)):
                tm.set_option(N.TONEMAPPER_OPTION_TABLE, mode)
                for _ in range(20):
                    tm.render()
                ctx.synchronize()
                n = 300
                t0 = time.perf_counter()
                for _ in range(n):
                    tm.render()
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / n
                rec[name] = {"us_per_pass": dt * 1e6, "GB_per_s": 12.0 * w * h / dt / 1e9}
            out["%s_%dx%d" % (kind, w, h)] = rec
            tm.destroy()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
