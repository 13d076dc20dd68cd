import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import vpt_amd
from vpt_amd.scene import Node, Transform, default_camera
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume
n, W, H, wide = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = vpt_amd.Context(0)
gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(n, noise=48.0), 'linear')
gvol.set_wide_tables(wide)
camera = default_camera(W / H); transform = Transform(Node())
def renderer(**opts):
    o = {'resolution': (W, H), 'transform': transform, 'rng': GoldenRatioRng()}; o.update(opts)
    return vpt_amd.MCMRenderer(ctx, gvol, camera, None, o)
def timed(r, frames, prof):
    r.reset()
    for _ in range(10): r.render()
    ctx.synchronize()
    r.set_profiling(prof)
    t0 = time.perf_counter()
    for _ in range(frames): r.render()
    t1 = time.perf_counter()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ms, l = r.profile(); r.set_profiling(False)
    return "wall %.4f ms/frame (enqueue %.4f) kernel %.4f" % (dt / frames * 1e3, (t1 - t0) / frames * 1e3, ms / max(l, 1))
for name, opts in (("shard 3/8", {'shard': (3, 8, 8)}), ("full", {}), ("shard 0/2", {'shard': (0, 2, 8)})):
    r = renderer(**opts)
    for prof in (4, 0):
        print(name, "prof", prof, timed(r, 200, prof), flush=True)
    r.destroy()
