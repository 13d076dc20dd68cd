#!/usr/bin/env python3
"""Times the BASELINE.json configs C2-C4 (and H) on one GPU: per-renderer ms/frame, samples/s, kernel time.
Not the headline bench (that is bench.py); used to fill BASELINE.md §5 and to catch scale problems early."""
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vpt_amd
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng


def run(kind, n, w, h, frames, warmup, **props):
    ctx = vpt_amd.Context(0)
    t0 = time.time()
    vol = sphere_volume(n, noise=48.0 if kind in ('mcs', 'mcm') else 0.0)
    t_gen = time.time() - t0
    t0 = time.time()
    gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
    ctx.synchronize()
    t_up = time.time() - t0
    r = vpt_amd.RendererFactory(kind)(ctx, gvol, default_camera(w / h), None,
                                      {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    optm = props.pop('_mcm_persistent', None)
    if optm is not None:
        r.set_option(1, optm)
        props = dict(props, mcm_persistent=optm)
    opt = props.pop('_mcs_persistent', None)
    if opt is not None:
        r.set_option(0, opt)
        props = dict(props, mcs_persistent=opt)
    for k, v in props.items():
        if k not in ('mcs_persistent', 'mcm_persistent'):
            setattr(r, k, v)
    r.reset()
    for _ in range(warmup):
        r.render()
    ctx.synchronize()
    r.clear_sample_count(); r.set_profiling(True)
    t0 = time.time()
    for _ in range(frames):
        r.render()
    ctx.synchronize()
    dt = time.time() - t0
    ms, nl = r.profile()
    samples = r.sample_count()
    out = {"renderer": kind, "volume": n, "size": [w, h], "frames": frames, "props": props,
           "ms_per_frame": dt / frames * 1e3, "kernel_ms": ms / max(nl, 1), "samples_per_frame": samples / frames,
           "samples_per_s": samples / dt, "volume_gen_s": t_gen, "upload_brick_s": t_up, "bricked_MiB": gvol.bricked_bytes() / 2 ** 20}
    print(json.dumps(out), flush=True)
    r.destroy(); gvol.destroy(); ctx.destroy()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3", "h", "c4"]
    if "c1" in which: run('mip', 64, 256, 256, 100, 10)
    if "mip" in which: run('mip', 256, 1920, 1080, 100, 10)
    if "c2" in which: run('eam', 256, 1920, 1080, 100, 10)
    if "c3" in which: run('mcs', 512, 1920, 1080, 256, 10)
    if "h" in which: run('mcm', 512, 1920, 1080, 100, 10)
    if "c4" in which: run('mcm', 1024, 1920, 1080, 100, 10)
    if "c5" in which: run('mcm', 2048, 3840, 2160, 20, 3)
    if "mcs_ext" in which:
        for ext in (1, 10, 50, 200):
            for pers in (0, 1):
                run('mcs', 512, 1920, 1080, 60, 6, extinction=ext, _mcs_persistent=pers)
    if "mcm_ext" in which:
        for ext in (1, 10, 50, 200):
            run('mcm', 512, 1920, 1080, 60, 6, extinction=ext)
    if "mcm_steps" in which:
        for pers in (0, 1):
            for st in (1, 2, 4, 8, 16, 32):
                run('mcm', 512, 1920, 1080, 60, 6, steps=st, _mcm_persistent=pers)
    for w in which:
        if w.startswith("mcm_one_"):
            run('mcm', 512, 1920, 1080, 60, 6, steps=int(w.split("_")[-1]), _mcm_persistent=0)
