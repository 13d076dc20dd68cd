#!/usr/bin/env python3
"""DOS renderer: time of one full sweep (reset + ceil(slices/steps) render() calls) at 1920x1080, per slice and per sweep.
A reference-shaped pass moves 40 B per pixel and slice over the WHOLE image (colour 16 B read + 16 B written, occlusion
4 B read + 4 B written); the native pass only launches the tiles of the volume's screen bounding box and updates the
colour in place, so `whole_image_equivalent_GB_per_s` (40 B x W x H per slice) is a comparison figure, not traffic.
Usage: python tools/dos_rate.py [--volume 256] [--slices 200] [--samples 8] [--sweeps 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--slices", type=int, default=200)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--sweeps", type=int, default=5)
    ap.add_argument("--filter", default="linear", choices=["linear", "nearest"])
    ap.add_argument("--extinction", type=float, default=100.0)
    args = ap.parse_args()
    import vpt_amd
    from vpt_amd.scene import default_camera, Transform, Node
    from vpt_amd.synthetic import sphere_volume, GoldenRatioRng
    W, H = args.width, args.height
    ctx = vpt_amd.Context(0)
    vol = vpt_amd.Volume.from_array(ctx, sphere_volume(args.volume, noise=40.0), args.filter)
    r = vpt_amd.DOSRenderer(ctx, vol, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    r.slices = args.slices; r.steps = args.steps; r.samples = args.samples; r.extinction = args.extinction
    r.generateOcclusionSamples()

    def sweep():
        r.reset()
        n = 0
        while True:
            r.render()
            if len(r._slices) == 0:
                return n
            n += len(r._slices)

    sweep(); ctx.synchronize()
    t0 = time.perf_counter()
    total = 0
    for _ in range(args.sweeps):
        total += sweep()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / args.sweeps
    per_slice = dt / (total / args.sweeps)
    ns = r.sample_count() / (args.sweeps + 1)
    print(json.dumps({"renderer": "dos", "volume": args.volume, "image": [W, H], "slices_per_sweep": total // args.sweeps, "occlusion_samples": args.samples,
                      "ms_per_sweep": dt * 1e3, "us_per_slice": per_slice * 1e6, "volume_samples_per_sweep": ns,
                      "whole_image_equivalent_GB_per_s": 40.0 * W * H / per_slice / 1e9}))
    r.destroy(); vol.destroy(); ctx.destroy()


if __name__ == "__main__":
    main()
