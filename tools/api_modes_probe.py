#!/usr/bin/env python3
"""What the API's modes cost with the library's defaults (profiles/experiments.md, round 4): microseconds per frame of
  * the render() loop, eager / fused / every-frame-written sequences by passes per call (vpt_renderer_play), and VPT_PLAY_GRAPH,
  * render() as the reference's three hooks (`fused: False`) against the fused call,
  * frame sizes from 256^2 to 2160p
on a synthetic 512^3 volume.  python tools/api_modes_probe.py [play|hooks|sizes] ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

what = sys.argv[1:] or ["play", "hooks", "sizes"]
cache = "/tmp/vpt_vol_512.npy"
vol = np.load(cache) if os.path.exists(cache) else sphere_volume(512, noise=48.0)
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')


def renderer(kind, W, H, **opts):
    o = {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
    o.update(opts)
    r = vpt_amd.RendererFactory(kind)(ctx, gvol, default_camera(W / H), None, o)
    r.reset()
    return r


def timed(go, frames=320):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        go(64); ctx.synchronize()
    ts = []
    for _ in range(3):
        ctx.synchronize(); t0 = time.perf_counter(); go(frames); ctx.synchronize()
        ts.append((time.perf_counter() - t0) / frames * 1e6)
    return sorted(ts)[1]


if "play" in what:
    for kind in ('mip', 'eam', 'iso', 'depth', 'mcs', 'mcm'):
        for fast in ((0, 1) if kind == 'mcm' else (0,)):
            for mode in ('render', 'eager4', 'eager16', 'graph16', 'fused2', 'fused4', 'fused8', 'fused16') + (('frames4', 'frames16') if kind == 'mcm' else ()):
                r = renderer(kind, 1920, 1080)
                if kind == 'mcm':
                    r.set_option(N.OPTION_FAST_MATH, fast)
                k = int(''.join(c for c in mode if c.isdigit()) or 1)

                def go(n):
                    if mode == 'render':
                        for _ in range(n):
                            r.render()
                    else:
                        for _ in range(n // k):
                            r.play(k, use_graph=mode.startswith('graph'), fused=mode.startswith('fused'), frames=mode.startswith('frames'))
                print('%-5s fast %d %-8s %8.2f us per frame' % (kind, fast, mode, timed(go)), flush=True)
                r.destroy()
if "hooks" in what:
    for kind in ('mip', 'eam', 'mcs', 'iso', 'depth', 'mcm'):
        for fused in (True, False):
            r = renderer(kind, 1920, 1080, fused=fused)
            t = timed(lambda n: [r.render() for _ in range(n)], 300)
            print('%-5s %-28s %8.2f us per frame' % (kind, 'render() = one fused call' if fused else 'render() = the three hooks', t), flush=True)
            r.destroy()
if "sizes" in what:
    for (W, H) in ((256, 256), (512, 512), (1024, 1024), (1920, 1080), (3840, 2160)):
        for kind in ('mcm', 'eam', 'mip', 'mcs', 'iso', 'depth'):
            r = renderer(kind, W, H)
            t = timed(lambda n: [r.render() for _ in range(n)], 200)
            print('%4dx%-4d %-5s %8.2f us per frame' % (W, H, kind, t), flush=True)
            r.destroy()
gvol.destroy(); ctx.destroy()
