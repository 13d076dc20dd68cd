#!/usr/bin/env python3
"""What ONE rank of the 8-GPU headline run does per frame, measured on one GPU:
  * the MCM pass of rank 3's share of the 1920x1080 frame (interleaved 8-row blocks, shard = (3, 8, 8)), bit-exact and fast-math;
  * the event hand-off of the native gather pipeline on a frame of the same size (1920 x 136 rows, one-rank communicator: kernel +
    stream events + self-gather, against the plain render() of the same frame).
>= 6x at 8 GPUs on H needs kernel + hand-off <= single-GPU frame / 6.  Writes one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd import _native as N                                   # noqa: E402
from vpt_amd.scene import Node, Transform, default_camera          # noqa: E402
from vpt_amd.synthetic import GoldenRatioRng, sphere_volume        # noqa: E402
from vpt_amd.tiles import RcclFrameGather                          # noqa: E402


def timed(ctx, step, frames=400, reps=5):
    for _ in range(50):
        step()
    ctx.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / frames
        best = dt if best is None else min(best, dt)
    return best * 1e6


def make(ctx, gvol, W, H, fast, split, classes=1, shard=None):
    o = {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
    if shard:
        o['shard'] = shard
    r = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, o)
    r.set_option(N.OPTION_FAST_MATH, fast)
    r.set_option(N.OPTION_TILE_CLASSES, classes)
    if split > 1:
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
    r.reset()
    return r


def main():
    W, H = 1920, 1080
    ctx = vpt_amd.Context(0)
    gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
    out = {"unit": "us per frame", "frame": "%dx%d" % (W, H), "volume": "512^3",
           "forms": "tile classes in force unless named general_kernel; N streams = VPT_OPTION_SPLIT_STREAMS (classes: HIT tiles | MISS tiles)"}
    for fast in ((1, 0) if os.environ.get('VPT_PROBE_FAST_FIRST') else (0, 1)):
        tag = "fast_math" if fast else "bit_exact"
        full = make(ctx, gvol, W, H, fast, 2)
        out["full_frame_two_streams_%s" % tag] = timed(ctx, full.render, 200)         # the N = 1 default of bench.py
        out["full_frame_tiles_%s" % tag] = full.tile_classes()[:2]
        full.destroy()
        full = make(ctx, gvol, W, H, fast, 3, classes=0)
        out["full_frame_general_kernel_three_streams_%s" % tag] = timed(ctx, full.render, 200)   # round 2's form
        full.destroy()
        for world, rank in (() if os.environ.get('VPT_PROBE_HANDOFF_ONLY') else ((8, 3), (4, 1), (2, 0))):
            for split in (1, 2, 3):
                sh = make(ctx, gvol, W, H, fast, split, shard=(rank, world, 8))
                out["shard_%d_of_%d_%d_streams_%s" % (rank, world, split, tag)] = timed(ctx, sh.render)
                if split == 1:
                    out["shard_%d_of_%d_tiles" % (rank, world)] = sh.tile_classes()[:2]
                    out["shard_%d_of_%d_rows" % (rank, world)] = int(sh.local_rows())
                sh.destroy()
            sh = make(ctx, gvol, W, H, fast, 2, classes=0, shard=(rank, world, 8))
            out["shard_%d_of_%d_general_kernel_two_streams_%s" % (rank, world, tag)] = timed(ctx, sh.render)
            sh.destroy()
        # VPT_OPTION_BUCKET_KERNEL: buckets of F frames through vpt_renderer_play_into (the renderer's frame ring stands in for the bucket)
        import ctypes as C
        for world, rank in (() if os.environ.get('VPT_PROBE_HANDOFF_ONLY') else ((8, 3), (4, 1), (2, 0), (1, 0))):
            for bucket in (0, 1):
                sh = make(ctx, gvol, W, H, fast, 2, shard=(rank, world, 8) if world > 1 else None)
                sh.set_option(N.OPTION_BUCKET_KERNEL, bucket)
                sh.play(16, frames=True)
                p, n = C.c_void_p(), C.c_size_t()
                N.check(N.lib().vpt_renderer_frame_ring_device(sh._h, C.byref(p), C.byref(n)))
                for F in (8, 16):
                    t = timed(ctx, lambda: sh.play_into(F, p.value, n.value), 50) / F
                    name = "shard_%d_of_%d" % (rank, world) if world > 1 else "full_frame"
                    out["%s_buckets_of_%d_%s_%s" % (name, F, "bucket_kernel" if bucket else "frame_by_frame", tag)] = t
                sh.set_render_target(0, 0)
                sh.destroy()
        # a full frame of a 1/8 shard's size through the gather pipeline (one-rank communicator) against its plain render()
        hs = 136
        plain = make(ctx, gvol, W, hs, fast, 2)
        t_plain = timed(ctx, plain.render)
        piped = make(ctx, gvol, W, hs, fast, 2, shard=(0, 1, 8))
        g = RcclFrameGather(piped, RcclFrameGather.unique_id(), 0, 1)
        for root in (0, -1):
            g.set_root(root)
            t_pipe = timed(ctx, g.render)
            g.synchronize()
            out["handoff_root%s_%s" % ("0" if root == 0 else "_all", tag)] = t_pipe - t_plain
        out["frame_1920x136_plain_two_streams_%s" % tag] = t_plain
        g.destroy(); piped.destroy(); plain.destroy()
        one = out["full_frame_two_streams_%s" % tag]
        for world, rank in (() if os.environ.get('VPT_PROBE_HANDOFF_ONLY') else ((8, 3), (4, 1), (2, 0))):
            k = min(out["shard_%d_of_%d_%d_streams_%s" % (rank, world, sp, tag)] for sp in (1, 2, 3))
            hnd = max(out["handoff_root0_%s" % tag], 0.0)
            out["per_rank_frame_at_%d_%s" % (world, tag)] = k + hnd
            out["projected_speedup_at_%d_%s" % (world, tag)] = one / (k + hnd)
            kb = out["shard_%d_of_%d_buckets_of_8_bucket_kernel_%s" % (rank, world, tag)]
            out["projected_speedup_at_%d_bucket_kernel_%s" % (world, tag)] = {
                "per_rank_frame": kb + hnd, "against_the_single_gpu_frame_launched_frame_by_frame": one / (kb + hnd),
                "against_the_single_gpu_frame_in_buckets_too": out["full_frame_buckets_of_8_bucket_kernel_%s" % tag] / (kb + hnd)}
    out["note"] = ("projected = this round's single-GPU frame (tile classes, two streams) / (best one-GPU time of one rank's share + hand-off); "
                   "UNMEASURED on more than one GPU.  bucket_kernel = VPT_OPTION_BUCKET_KERNEL (8 frames of a bucket by one launch per tile class; every "
                   "frame rendered and written); the frame gather itself is bound by xGMI (2.07 MB per link and 1080p RGBA16F frame: ~14 us at 153 GB/s), "
                   "which no projection here includes.  hand-off = one frame through the native pipeline on a one-rank communicator minus the plain render() of the same frame, "
                   "both one call per frame from Python: a difference of two host-paced 18-23 us measurements, 1.5-6.2 us from box to box; 16 frames per "
                   "vpt_gather_play call are SLOWER (29-38 us per frame: with the host far ahead every stream wait becomes a barrier packet)")
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
