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


def _drive(anim, clock, step):
    op = step[0]
    if op == 'rotate':
        anim._rotateAroundFocus(step[1], step[2])
    elif op == 'zoom':
        anim._zoom(step[1])
    elif op == 'move':
        anim._move(list(step[1]))
    elif op == 'pointerdown':
        anim._handlePointerDown({'button': step[1], 'pointerId': 1})
    elif op == 'pointerup':
        anim._handlePointerUp({'pointerId': 1})
    elif op == 'pointermove':
        anim._handlePointerMove({'movementX': step[1], 'movementY': step[2], 'shiftKey': step[3]})
    elif op == 'wheel':
        anim._handleWheel({'deltaY': step[1]})
    elif op == 'keydown':
        anim._handleKeyDown({'key': step[1]})
    elif op == 'keyup':
        anim._handleKeyUp({'key': step[1]})
    elif op == 'tick':
        clock[0] += step[1]
        anim._update()


def test_orbit_camera_animator_matches_reference_bits():
    """the reference's OrbitCameraAnimator.js run under node on scripted input (tests/golden/run_reference_orbit.mjs):
    every transform it assigns, and the TypeError of its middle-button drag"""
    from vpt_amd.animators import OrbitCameraAnimator
    d = json.load(open(os.path.join(HERE, "golden", "orbit_animator_r01.json")))
    assert len(d["cases"]) == 3
    checked = 0
    for c in d["cases"]:
        node = Node()
        node.transform.localTranslation = c["start"]
        clock = [1000]
        a = OrbitCameraAnimator(node, None, dict(c["options"], now=lambda: clock[0]))
        for f in c["frames"]:
            v0 = node.transform.version
            thrown = None
            try:
                _drive(a, clock, f["step"])
            except TypeError:
                thrown = "TypeError"
            assert thrown == f["throws"], f["step"]
            if f["translation_bits"] is None:
                assert node.transform.version == v0, f["step"]          # the reference assigned nothing either
            else:
                assert node.transform.localTranslation.view(np.uint32).tolist() == f["translation_bits"], f["step"]
                assert node.transform.localRotation.view(np.uint32).tolist() == f["rotation_bits"], f["step"]
                checked += 1
    assert checked >= 15
