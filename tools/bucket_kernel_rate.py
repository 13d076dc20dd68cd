#!/usr/bin/env python3
"""One rank's share of the headline frame (MCM 512^3 @ 1920x1080, default camera) as buckets of F frames through
vpt_renderer_play_into — the compute side of bench.py's torch.distributed pipeline — frame by frame and with
VPT_OPTION_BUCKET_KERNEL (one launch per tile class and bucket).  us per frame; the renderer's frame ring stands in for the
collective's bucket.  Usage: bucket_kernel_rate.py [fast-math 0|1]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

fast = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cache = "/tmp/vpt_vol_512.npy"
if os.path.exists(cache):
    vol = np.load(cache)
else:
    vol = sphere_volume(512, noise=48.0); np.save(cache, vol)
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
W, H = 1920, 1080
for shard in ((3, 8, 8), (1, 4, 8), (0, 2, 8), None):
    for bucket in (0, 1):
        opts = {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
        if shard:
            opts['shard'] = shard
        r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, opts)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
       
        r.set_option(N.OPTION_BUCKET_KERNEL, bucket)
        r.reset()
        r.play(16, frames=True)
        p, n = C.c_void_p(), C.c_size_t()
        N.check(N.lib().vpt_renderer_frame_ring_device(r._h, C.byref(p), C.byref(n)))
        for F in (4, 8, 16):
            for _ in range(20):
                r.play_into(F, p.value, n.value)
            ctx.synchronize()
            best = 1e9
            for _ in range(5):
                r.join(); ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    r.play_into(F, p.value, n.value)
                r.join(); ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / (40 * F) * 1e6)
            print("fast-math %d  share %-10s bucket kernel %d  F = %2d: %6.2f us per frame%s" % (
                fast, "%d of %d" % (shard[0], shard[1]) if shard else "whole", bucket, F, best,
                "  (%d bucket launches)" % r.bucket_launches() if bucket else ""), flush=True)
        r.set_render_target(0, 0)
        r.destroy()
