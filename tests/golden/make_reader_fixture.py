#!/usr/bin/env python3
"""Generates tests/golden/readers_r01.json by RUNNING THE REFERENCE'S OWN READERS (RAWReader.js, ZIPReader.js,
BVPReader.js, ReaderFactory.js — imported in place by node through tests/golden/esm_loader.mjs, nothing copied) over a
small synthetic BVP archive and a raw volume built here.  The fixture holds the inputs (base64) and what the reference
returned (metadata objects, central-directory entries, per-file lengths and SHA-256 digests, error messages).
Build container only (needs /root/reference and node); run from the repository root:
    python tests/golden/make_reader_fixture.py"""
import base64
import io
import json
import os
import subprocess
import tempfile
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_archive():
    """an 8x6x5 u8 volume cut into three blocks with different shapes and positions, stored (method 0) as the format requires"""
    rng = np.random.Generator(np.random.PCG64(77))
    vol = rng.integers(0, 256, size=(5, 6, 8), dtype=np.uint8)           # [z][y][x]
    blocks = [
        {"name": "blocks/a.raw", "pos": (0, 0, 0), "dim": (8, 6, 2)},        # two full slices
        {"name": "blocks/b.raw", "pos": (0, 0, 2), "dim": (5, 6, 3)},        # left part of the upper three slices
        {"name": "blocks/c.raw", "pos": (5, 0, 2), "dim": (3, 6, 3)},        # right part
    ]
    manifest = {
        "meta": {"version": 1},
        "modalities": [{
            "name": "default",
            "dimensions": {"width": 8, "height": 6, "depth": 5},
            "transform": {"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]},
            "format": 6403, "internalFormat": 33321, "type": 5121,
            "placements": [{"index": i, "position": {"x": b["pos"][0], "y": b["pos"][1], "z": b["pos"][2]}} for i, b in enumerate(blocks)],
        }, {
            "name": "second", "dimensions": {"width": 8, "height": 6, "depth": 2},
            "transform": {"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]},
            "format": 6403, "internalFormat": 33321, "type": 5121,
            "placements": [{"index": 0, "position": {"x": 0, "y": 0, "z": 0}}],
        }],
        "blocks": [{"url": b["name"], "format": "raw", "dimensions": {"width": b["dim"][0], "height": b["dim"][1], "depth": b["dim"][2]}} for b in blocks],
    }
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", compression=zipfile.ZIP_STORED) as z:
        for i, b in enumerate(blocks):
            x, y, zz = b["pos"]; w, h, d = b["dim"]
            info = zipfile.ZipInfo(b["name"], date_time=(2024, 1, 1, 0, 0, 0))
            if i == 1:
                info.comment = b"entry comment"                              # exercises fileCommentLength in the central directory
                info.extra = b"\xca\xfe\x04\x00abcd"                         # and an extra field in both headers
            z.writestr(info, np.ascontiguousarray(vol[zz:zz + d, y:y + h, x:x + w]).tobytes())
        z.writestr(zipfile.ZipInfo("manifest.json", date_time=(2024, 1, 1, 0, 0, 0)), json.dumps(manifest))
    return bio.getvalue(), vol


def main():
    archive, vol = synthetic_archive()
    raw = np.random.Generator(np.random.PCG64(78)).integers(0, 256, size=(3, 4, 5), dtype=np.uint8)
    with tempfile.TemporaryDirectory(dir=HERE) as tmp:
        ap, rp = os.path.join(tmp, "a.bvp"), os.path.join(tmp, "v.raw")
        open(ap, "wb").write(archive); open(rp, "wb").write(raw.tobytes())
        res = subprocess.run(["node", "--no-warnings", "--experimental-loader", os.path.join(HERE, "esm_loader.mjs"),
                              os.path.join(HERE, "run_reference_readers.mjs"), ap, rp, "5", "4", "3"],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=False)
    if res.returncode != 0:
        raise SystemExit("node failed:\n" + res.stderr.decode())
    ref = json.loads(res.stdout.decode())
    out = {
        "_comment": "outputs of the reference's readers run in the build container (tests/golden/make_reader_fixture.py); inputs synthetic",
        "archive_base64": base64.b64encode(archive).decode(), "archive_volume_zyx": vol.reshape(-1).tolist(), "archive_dims_whd": [8, 6, 5],
        "raw_base64": base64.b64encode(raw.tobytes()).decode(), "raw_dims_whd": [5, 4, 3],
        "reference": ref,
    }
    path = os.path.join(HERE, "readers_r01.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes; files:", ref["zip_files"])


if __name__ == "__main__":
    main()
