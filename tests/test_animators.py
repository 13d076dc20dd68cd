"""CPU: CircleAnimator against the reference's own CircleAnimator.js run under node
(tests/golden/circle_animator_r01.json, made by tests/golden/run_reference_animator.mjs), and the PNG encoder."""
import json
import os

import numpy as np

from vpt_amd.animators import CircleAnimator
from vpt_amd.png import encode_png, decode_png
from vpt_amd.scene import Node

HERE = os.path.dirname(os.path.abspath(__file__))


def test_circle_animator_matches_reference_bits():
    d = json.load(open(os.path.join(HERE, "golden", "circle_animator_r01.json")))
    assert len(d["cases"]) == 4
    for c in d["cases"]:
        node = Node()
        hits = []
        node.transform.addEventListener('change', lambda e: hits.append(1))
        a = CircleAnimator(node, c["options"])
        for f in c["frames"]:
            a.update(f["t"])
            assert node.transform.localTranslation.view(np.uint32).tolist() == f["translation_bits"], (c["options"], f["t"])
            assert node.transform.localRotation.view(np.uint32).tolist() == f["rotation_bits"], (c["options"], f["t"])
        assert len(hits) == 2 * len(c["frames"])          # both setters fire 'change' (RenderingContext.js:42-46 resets on it)


def test_png_round_trip():
    img = np.random.default_rng(0).integers(0, 256, size=(9, 13, 4), dtype=np.uint8)
    data = encode_png(img)
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[-12:-8] == b"\x00\x00\x00\x00" and data[-8:-4] == b"IEND"
    assert (decode_png(data) == img[::-1]).all()           # GL row 0 (bottom) is the PNG's last row
    assert (decode_png(encode_png(img, bottom_up=False)) == img).all()
