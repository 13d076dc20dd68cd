#!/usr/bin/env python3
"""Time of the GPU re-layout of an uploaded volume into apron bricks (vpt_volume_finalize -> k_brickify), per size.
Bytes moved: n^3 read + 2 n^3 written (125 of every 128 bytes are voxels and apron)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd import _native as N                                   # noqa: E402


def main():
    ctx = vpt_amd.Context(0)
    out = {}
    for n in (128, 256, 512, 1024):
        vol = np.random.default_rng(n).integers(0, 256, size=(n, n, n), dtype=np.uint8)
        g = vpt_amd.Volume.from_array(ctx, vol, 'linear')
        L = N.lib()
        blk = np.ascontiguousarray(vol[:1, :1, :1])
        times = []
        for _ in range(6):
            N.check(L.vpt_volume_upload_block(g.getTexture(), 0, 0, 0, 1, 1, 1, blk.ctypes.data_as(C.c_void_p), 1))   # marks the volume dirty
            ctx.synchronize()
            t0 = time.perf_counter()
            N.check(L.vpt_volume_finalize(g.getTexture()))
            ctx.synchronize()
            times.append(time.perf_counter() - t0)
        dt = min(times[1:])
        out["%d^3" % n] = {"us": dt * 1e6, "GB_per_s": 3.0 * n ** 3 / dt / 1e9}
        g.destroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
