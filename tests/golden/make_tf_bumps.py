#!/usr/bin/env python3
"""Writes tests/golden/tf_bumps_r04.json: three bump files in the reference's Save format (TransferFunction.js:82-84: the bump array)
with the SHA-256 of the texels oracle/vpt_tonemap_oracle.c vpo_tf_rasterize makes of them (256 x 256 and 64 x 3, un-premultiplied and
premultiplied).  The digests pin the restatement against drift; the HIP rasteriser and both hosts are compared with them.
Parity unpinned: the reference holds no rendered transfer function to compare with."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

FILES = {
    "default_bump": [{"position": {"x": 0.5, "y": 0.5}, "size": {"x": 0.2, "y": 0.2}, "color": {"r": 1, "g": 0, "b": 0, "a": 1}}],
    "three_overlapping": [
        {"position": {"x": 0.25, "y": 0.9}, "size": {"x": 0.1, "y": 0.6}, "color": {"r": 0.9, "g": 0.3, "b": 0.1, "a": 0.4}},
        {"position": {"x": 0.5, "y": 0.5}, "size": {"x": 0.3, "y": 0.05}, "color": {"r": 0.1, "g": 0.8, "b": 0.2, "a": 0.9}},
        {"position": {"x": 0.62, "y": 0.55}, "size": {"x": 0.08, "y": 0.4}, "color": {"r": 0.2, "g": 0.2, "b": 1.0, "a": 0.65}}],
    "edges_and_excess": [
        {"position": {"x": 0.0, "y": 1.0}, "size": {"x": 0.5, "y": 0.5}, "color": {"r": 2.0, "g": 0.5, "b": -0.5, "a": 1.5}},
        {"position": {"x": 1.0, "y": 0.0}, "size": {"x": 0.02, "y": 3.0}, "color": {"r": 0.3, "g": 0.3, "b": 0.3, "a": 0.3}}],
}


def pack(bumps):
    return np.array([[b["position"]["x"], b["position"]["y"], b["size"]["x"], b["size"]["y"],
                      b["color"]["r"], b["color"]["g"], b["color"]["b"], b["color"]["a"]] for b in bumps], dtype=np.float32).reshape(-1, 8)


def main():
    out = {"_what": __doc__.strip().split("\n\n")[0].replace("\n", " "), "files": {}}
    for name, bumps in FILES.items():
        e = {"bumps": bumps, "sha256": {}}
        for (w, h) in ((256, 256), (64, 3)):
            for un in (1, 0):
                t = O.tf_rasterize(pack(bumps), w, h, bool(un))
                e["sha256"]["%dx%d_%s" % (w, h, "unpremultiplied" if un else "premultiplied")] = hashlib.sha256(t.tobytes()).hexdigest()
        out["files"][name] = e
    with open(os.path.join(ROOT, "tests", "golden", "tf_bumps_r04.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["files"]), "bump files")


if __name__ == "__main__":
    main()
