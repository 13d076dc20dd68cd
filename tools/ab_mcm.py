#!/usr/bin/env python3
"""One build of the library, one configuration of the headline workload (MCM 512^3 @ 1920x1080, default camera), timed the way
bench.py times it (blocks of N render() calls between synchronisations, median block) but without the oracle frame check —
the A/B workhorse of tools/ab.sh.  The library is chosen through VPT_HIP_LIBRARY (vpt_amd/_native.py), the in-tree artefact is
not touched.  Prints one line: tag, median / min us per frame."""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--tag", default="")
ap.add_argument("--fast", type=int, default=1)
ap.add_argument("--split", type=int, default=0, help="VPT_OPTION_SPLIT_STREAMS; 0 = the library's default")
ap.add_argument("--classes", type=int, default=1, help="VPT_OPTION_TILE_CLASSES: 0 off, 1 on, 2 on even on one stream (each class kernel alone on the chip)")
ap.add_argument("--volume", type=int, default=512)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--blocks", type=int, default=5)
ap.add_argument("--shard", default="", help="rank,world,rows: time one rank's share of the frame")
ap.add_argument("--renderer", default="mcm")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--hit-form", type=int, default=0, help="1 / 2: force a form of the HIT-tile kernel (VPT_HIT_KERNEL_FORM in the environment of the renderer's creation)")
ap.add_argument("--persistent", type=int, default=0, help="VPT_OPTION_MCS_PERSISTENT / VPT_OPTION_MCM_PERSISTENT (the renderer's): the persistent-wave kernel forms")
ap.add_argument("--records", type=int, default=-1, help="VPT_OPTION_COLUMN_RECORDS (MCM): 0 / 1; -1 = the library's default")
ap.add_argument("--camera-z", type=float, default=2.0, help="z of the default camera's translation (0.9: the volume fills the frame, every tile is a HIT tile)")
ap.add_argument("--extinction", type=float, default=0.0)
ap.add_argument("--tf", default="", help="'ramp': the 256x1 grey ramp with alpha = v instead of the default 2x1 transfer function")
ap.add_argument("--kernel-times", type=int, default=0, help="1: HIP events around every 4th launch: per-kernel averages (context stream | side stream)")
ap.add_argument("--digest", type=int, default=0, help="1: print a digest of the radiance buffer after the run (same seeds: equal across bit-identical builds / options)")
ap.add_argument("--format", default="r8", help="r8 | rg8 | f32 | rg32f: the volume's texel format (the synthetic bytes, second channel = 255 - v)")
ap.add_argument("--play-fused", type=int, default=0, help="n > 0: time vpt_renderer_play(n, VPT_PLAY_FUSED) per pass instead of render()")
ap.add_argument("--torch-stream", type=int, default=0, help="1: the context runs on a torch.cuda.Stream, as in bench.py")
ap.add_argument("--dummy-contexts", type=int, default=0, help="contexts (one HIP stream each) created and used BEFORE the measured one: shifts which hardware queue each later stream lands on")
args = ap.parse_args()
if args.lib:
    os.environ["VPT_HIP_LIBRARY"] = os.path.abspath(args.lib)
if args.hit_form:
    os.environ["VPT_HIT_KERNEL_FORM"] = str(args.hit_form)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

cache = "/tmp/vpt_vol_%d.npy" % args.volume
if os.path.exists(cache):
    vol = np.load(cache)
else:
    from concurrent.futures import ThreadPoolExecutor
    n = args.volume
    vol = np.empty((n, n, n), dtype=np.uint8)

    def _slab(z0):
        vol[z0:z0 + 16] = sphere_volume(n, noise=48.0, z_range=(z0, min(n, z0 + 16)))
    with ThreadPoolExecutor(max_workers=min(14, len(os.sched_getaffinity(0)))) as ex:
        list(ex.map(_slab, range(0, n, 16)))
    np.save(cache, vol)
dummies = [vpt_amd.Context(0) for _ in range(args.dummy_contexts)]
for d in dummies:
    d.stream_read_rate(1 << 20, 1)                  # a launch on the stream: the queue is bound at first use
if args.torch_stream:
    import torch
    torch.cuda.set_device(0)
    _ts = torch.cuda.Stream(device=torch.device('cuda', 0))
    ctx = vpt_amd.Context(0, stream=_ts.cuda_stream)
else:
    ctx = vpt_amd.Context(0)
if args.format in ("rg8", "rg32f"):
    vol = np.stack([vol, 255 - vol], axis=-1)
if args.format in ("f32", "rg32f"):
    vol = vol.astype(np.float32) / 255.0
gvol = vpt_amd.Volume.from_array(ctx, vol, 'linear')
W, H = args.width, args.height
opts = {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
if args.shard:
    opts['shard'] = tuple(int(x) for x in args.shard.split(","))
camera = default_camera(W / H)
if args.camera_z != 2.0:
    camera.transform.localTranslation = [0.0, 0.0, args.camera_z]
r = vpt_amd.RendererFactory(args.renderer)(ctx, gvol, camera, None, opts)
if args.extinction:
    r.extinction = args.extinction
if args.tf == "ramp":
    from vpt_amd.synthetic import ramp_tf
    r.setTransferFunction(ramp_tf(256))
if args.renderer == "mcm":
    r.set_option(N.OPTION_FAST_MATH, args.fast)
    r.steps = args.steps
try:
    r.set_option(N.OPTION_TILE_CLASSES, args.classes)
except vpt_amd.VptError:
    pass                                              # a build from before the option existed
if args.persistent and args.renderer in ("mcs", "mcm"):
    r.set_option(N.OPTION_MCS_PERSISTENT if args.renderer == "mcs" else N.OPTION_MCM_PERSISTENT, args.persistent)
if args.records >= 0 and args.renderer == "mcm":
    r.set_option(N.OPTION_COLUMN_RECORDS, args.records)
if args.split >= 1:                                   # 0: the library's own default
    r.set_option(N.OPTION_SPLIT_STREAMS, args.split)
r.reset()
def frames(n):
    if args.play_fused:
        for _ in range(max(1, n // args.play_fused)):
            r.play(args.play_fused, fused=True)
    else:
        for _ in range(n):
            r.render()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    frames(50)
    ctx.synchronize()
blocks = []
for _ in range(args.blocks):
    ctx.synchronize()
    t0 = time.perf_counter()
    frames(args.frames)
    ctx.synchronize()
    blocks.append((time.perf_counter() - t0) / (max(1, args.frames // args.play_fused) * args.play_fused if args.play_fused else args.frames) * 1e6)
blocks.sort()
med = blocks[len(blocks) // 2]
extra = ""
if args.kernel_times:
    r.set_profiling(4)
    for _ in range(200):
        r.render()
    ctx.synchronize()
    ms, n = r.profile(); ms2, n2 = r.profile_side()
    r.set_profiling(False)
    if args.split == 1 and n2:          # one stream, VPT_OPTION_TILE_CLASSES 2: HIT then MISS; the outer pair brackets both
        extra += "  kernels alone: HIT %.2f us | MISS %.2f us" % ((ms / max(n, 1) - ms2 / n2) * 1e3, ms2 / n2 * 1e3)
    else:
        extra += "  kernels: context %.2f us" % (ms / max(n, 1) * 1e3) + ((" | side %.2f us" % (ms2 / n2 * 1e3)) if n2 else "")
    if args.renderer == "mcm":
        extra += "  tiles %s" % (r.tile_classes()[:2],)
if args.digest:
    import hashlib
    extra += "  digest " + hashlib.sha256(r.read(N.BUFFER_MCM_RADIANCE if args.renderer == "mcm" else N.BUFFER_RENDER).tobytes()).hexdigest()[:12]
samples = W * H * args.steps if not args.shard else None
print("%-28s fast %d split %d classes %d%s: median %7.2f us  min %7.2f  max %7.2f%s" % (
    (args.tag or os.path.basename(args.lib) or "in-tree") + (" dummies %d" % args.dummy_contexts if args.dummy_contexts else "") + (" hit-form %d" % args.hit_form if args.hit_form else ""), args.fast, args.split, args.classes, (" shard " + args.shard) if args.shard else "",
    med, blocks[0], blocks[-1], ("  frac %.3f" % (24.0 * samples / (med * 1e-6) / 8e12)) if samples and args.renderer == "mcm" else "") +
    (" records %d" % args.records if args.records >= 0 else "") + (" persistent %d" % args.persistent if args.persistent else "") + extra, flush=True)
r.destroy(); gvol.destroy(); ctx.destroy()
