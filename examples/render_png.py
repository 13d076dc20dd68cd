#!/usr/bin/env python3
"""Renders a volume with any renderer / tone mapper through the headless RenderingContext and writes PNG files.

    python examples/render_png.py --renderer mcm --tonemapper artistic --frames 64 --out out.png
    python examples/render_png.py --volume data.bvp ...        (BVP container)   --volume data.raw --dims 256 256 256

Without --volume a synthetic 128^3 sphere with lattice noise is used.  PNG encoding is plain zlib (no imaging library)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpt_amd                                                     # noqa: E402
from vpt_amd.synthetic import sphere_volume, colour_tf, GoldenRatioRng   # noqa: E402
from vpt_amd.png import write_png                                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", default="")
    ap.add_argument("--dims", type=int, nargs=3, default=None, help="width height depth of a .raw volume")
    ap.add_argument("--renderer", default="mcm")
    ap.add_argument("--tonemapper", default="artistic")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--extinction", type=float, default=None)
    ap.add_argument("--tf", default="default", choices=["default", "colour"])
    ap.add_argument("--yaw", type=float, default=0.6)
    ap.add_argument("--pitch", type=float, default=-0.35)
    ap.add_argument("--out", default="frame.png")
    a = ap.parse_args()

    rc = vpt_amd.RenderingContext({'resolution': (a.width, a.height), 'rng': GoldenRatioRng()})
    rc.resize(a.width, a.height)
    if a.volume.endswith(".bvp"):
        reader = vpt_amd.BVPReader(vpt_amd.FileLoader(a.volume))
    elif a.volume:
        w, h, d = a.dims
        reader = vpt_amd.RAWReader(vpt_amd.FileLoader(a.volume), {'width': w, 'height': h, 'depth': d})
    else:
        reader = vpt_amd.RAWReader(sphere_volume(128, noise=48.0), {'width': 128, 'height': 128, 'depth': 128})
    rc.setVolume(reader)
    # orbit the camera a little so that three faces of the volume are visible
    import math
    from vpt_amd.scene import quat
    q = quat.multiply(quat.create(), quat.setAxisAngle(quat.create(), [0, 1, 0], a.yaw), quat.setAxisAngle(quat.create(), [1, 0, 0], a.pitch))
    rc.camera.transform.localRotation = q
    d = 1.7
    rc.camera.transform.localTranslation = [d * math.sin(a.yaw) * math.cos(a.pitch), -d * math.sin(a.pitch), d * math.cos(a.yaw) * math.cos(a.pitch)]
    rc.chooseRenderer(a.renderer)
    rc.chooseToneMapper(a.tonemapper)
    if a.tf == "colour":
        rc.renderer.setTransferFunction(colour_tf(256, 1))
    if a.extinction is not None and hasattr(rc.renderer, 'extinction'):
        rc.renderer.extinction = a.extinction
    rc.renderer.reset()
    for _ in range(a.frames):
        rc.render()
    write_png(a.out, rc.getFrame())
    print("wrote %s (%s, %s, %d frames, %d volume samples)" % (a.out, a.renderer, a.tonemapper, a.frames, rc.renderer.sample_count()))
    rc.destroy()


if __name__ == "__main__":
    main()
