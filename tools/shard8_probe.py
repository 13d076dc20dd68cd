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


def main():
    W, H = 1920, 1080
    ctx = vpt_amd.Context(0)
    gvol = vpt_amd.Volume.from_array(ctx, sphere_volume(512, noise=48.0), 'linear')
    out = {"unit": "us per frame", "frame": "%dx%d" % (W, H), "volume": "512^3"}
    for fast in (0, 1):
        tag = "fast_math" if fast else "bit_exact"
        full = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
        full.set_option(N.OPTION_FAST_MATH, fast); full.reset()
        out["full_frame_%s" % tag] = timed(ctx, full.render, 200)
        full.set_option(N.OPTION_SPLIT_STREAMS, 3)
        out["full_frame_three_streams_%s" % tag] = timed(ctx, full.render, 200)      # the N = 1 default of bench.py
        full.destroy()
        sh = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng(), 'shard': (3, 8, 8)})
        sh.set_option(N.OPTION_FAST_MATH, fast); sh.reset()
        out["shard_3_of_8_kernel_%s" % tag] = timed(ctx, sh.render)
        out["shard_rows"] = int(sh.local_rows())
        sh.destroy()
        # a full frame of a shard's size through the gather pipeline (one-rank communicator) against its plain render()
        hs = 136
        plain = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, hs), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
        plain.set_option(N.OPTION_FAST_MATH, fast); plain.reset()
        t_plain = timed(ctx, plain.render)
        piped = vpt_amd.MCMRenderer(ctx, gvol, default_camera(W / H), None, {'resolution': (W, hs), 'transform': Transform(Node()), 'rng': GoldenRatioRng(), 'shard': (0, 1, 8)})
        piped.set_option(N.OPTION_FAST_MATH, fast); piped.reset()
        g = RcclFrameGather(piped, RcclFrameGather.unique_id(), 0, 1)
        for root in (0, -1):
            g.set_root(root)
            t_pipe = timed(ctx, g.render)
            g.synchronize()
            out["handoff_root%s_%s" % ("0" if root == 0 else "_all", tag)] = t_pipe - t_plain
        out["frame_1920x136_plain_%s" % tag] = t_plain
        g.destroy(); piped.destroy(); plain.destroy()
        k, hnd = out["shard_3_of_8_kernel_%s" % tag], max(out["handoff_root0_%s" % tag], 0.0)
        out["per_rank_frame_%s" % tag] = k + hnd
        out["projected_speedup_at_8_%s" % tag] = out["full_frame_%s" % tag] / (k + hnd)
        out["projected_speedup_at_8_vs_three_streams_%s" % tag] = out["full_frame_three_streams_%s" % tag] / (k + hnd)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
