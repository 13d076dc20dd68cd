#!/usr/bin/env python3
"""Does running the frame as K independent row-shards on K HIP streams hide the launch ramp and the tail of the MCM pass?
Each shard's pass depends only on its own previous pass, so the K streams never synchronise with each other; the tail of one
shard's launch overlaps the body of the others.  Prints ms per full frame for K = 1 (the plain renderer), 2, 3, 4.
(Probe only: every context holds its own copy of the volume here.)"""
import json
import sys
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

W, H, NV = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 512
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 0
vol = sphere_volume(NV, noise=48.0)
dev = torch.device("cuda", 0)
res = {}
for K in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    ctxs = [vpt_amd.Context(0, stream=s.cuda_stream) for s in streams]
    vols = [vpt_amd.Volume.from_array(c, vol, 'linear') for c in ctxs]
    rs = []
    for k in range(K):
        o = {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
        if K > 1:
            o['shard'] = (k, K, 8)
        r = vpt_amd.MCMRenderer(ctxs[k], vols[k], default_camera(W / H), None, o)
        if fast:
            r.set_option(N.OPTION_FAST_MATH, 1)
        r.reset()
        rs.append(r)
    for _ in range(50):
        for r in rs:
            r.render()
    torch.cuda.synchronize()
    best = None
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            for r in rs:
                r.render()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        best = dt if best is None else min(best, dt)
    res["streams_%d" % K] = best * 1e6
    print("K=%d  %.2f us per frame" % (K, best * 1e6), flush=True)
    for r in rs:
        r.destroy()
    for v in vols:
        v.destroy()
    for c in ctxs:
        c.destroy()
print(json.dumps(res))
