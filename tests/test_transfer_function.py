"""The transfer-function widget as data (vpt_amd/transfer_function.py, oracle vpo_tf_rasterize; reference:
src/js/ui/TransferFunction/TransferFunction.js:74-85,110-144, src/glsl/TransferFunction.glsl:32-35).
CPU: the oracle's restatement against a float64 evaluation of the same formulas (+-1 LSB per bump drawn), against the committed digests,
and the orientation / blending properties the canvas has.  GPU: the HIP rasteriser against the oracle, bit for bit, through the Python host.
Parity unpinned: the reference holds no rendered transfer function (mediump exp and the browser's un-premultiplication are
implementation-defined); the digests pin this restatement against drift."""
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = json.load(open(os.path.join(ROOT, "tests", "golden", "tf_bumps_r04.json")))


def pack(bumps):
    return np.array([[b["position"]["x"], b["position"]["y"], b["size"]["x"], b["size"]["y"],
                      b["color"]["r"], b["color"]["g"], b["color"]["b"], b["color"]["a"]] for b in bumps], dtype=np.float32).reshape(-1, 8)


def float64_canvas(bumps, w, h):
    """the same drawing in float64 WITHOUT the 8-bit store between bumps: premultiplied [h][w][4] in [0, 1], row 0 = top"""
    u = (np.arange(w) + 0.5) / w
    v = ((h - 1 - np.arange(h)) + 0.5) / h
    U, V = np.meshgrid(u, v)
    dst = np.zeros((h, w, 4))
    for b in pack(bumps).astype(np.float64):
        r2 = ((b[0] - U) / b[2]) ** 2 + ((b[1] - V) / b[3]) ** 2
        src = np.clip(b[4:8][None, None, :] * np.exp(-r2)[:, :, None], 0.0, 1.0)
        dst = np.clip(src + dst * (1.0 - src[:, :, 3:4]), 0.0, 1.0)
    return dst


@pytest.mark.parametrize("name", sorted(FIXTURE["files"]))
def test_oracle_matches_the_committed_digests_and_a_float64_evaluation(oracle, name):
    entry = FIXTURE["files"][name]
    for key, want in entry["sha256"].items():
        size, form = key.split("_")
        w, h = (int(x) for x in size.split("x"))
        t = oracle.tf_rasterize(pack(entry["bumps"]), w, h, form == "unpremultiplied")
        assert hashlib.sha256(t.tobytes()).hexdigest() == want, key
        if form == "premultiplied":
            ref = float64_canvas(entry["bumps"], w, h) * 255.0
            # every bump drawn rounds the target once more: +-0.5 LSB each, carried through the later blends
            assert np.abs(t.astype(np.float64) - ref).max() <= 0.5 * len(entry["bumps"]) + 0.01, key


def test_orientation_blending_and_unpremultiplication(oracle):
    top = [{"position": {"x": 0.25, "y": 1.0}, "size": {"x": 0.1, "y": 0.1}, "color": {"r": 0, "g": 1, "b": 0, "a": 1}}]
    t = oracle.tf_rasterize(pack(top), 64, 32, False)
    j, i = np.unravel_index(np.argmax(t[:, :, 3]), t.shape[:2])
    assert j == 0 and i in (15, 16)                               # position.y = 1: the top of the widget = texel row 0 (texImage2D(canvas))
    assert (t[:, :, 0] == 0).all() and (t[:, :, 2] == 0).all()
    # an opaque bump drawn later hides what is under it at its centre (ONE, ONE_MINUS_SRC_ALPHA)
    two = [{"position": {"x": 0.5, "y": 0.5}, "size": {"x": 0.5, "y": 0.5}, "color": {"r": 1, "g": 0, "b": 0, "a": 1}},
           {"position": {"x": 0.5, "y": 0.5}, "size": {"x": 0.05, "y": 0.05}, "color": {"r": 0, "g": 0, "b": 1, "a": 1}}]
    t = oracle.tf_rasterize(pack(two), 65, 65, False)
    assert tuple(t[32, 32]) == (0, 0, 255, 255)
    assert t[32, 0, 0] > 0 and t[32, 0, 2] == 0
    # un-premultiplied: alpha is the Gaussian, the colour stays the bump's own colour wherever alpha survives the 8 bits
    one = FIXTURE["files"]["default_bump"]["bumps"]
    p = oracle.tf_rasterize(pack(one), 256, 256, False)
    q = oracle.tf_rasterize(pack(one), 256, 256, True)
    assert (p[:, :, 3] == q[:, :, 3]).all()
    lit = q[:, :, 3] > 0
    assert (q[:, :, 0][lit] == 255).all() and (q[:, :, 1][lit] == 0).all() and (q[~lit] == 0).all()
    assert (p[:, :, 0] == p[:, :, 3]).all()                       # premultiplied red = alpha for color (1, 0, 0, 1)
    # no bumps: the cleared canvas
    assert (oracle.tf_rasterize(np.zeros((0, 8), np.float32), 7, 5, True) == 0).all()


def test_bump_list_operations_and_json(tmp_path):
    from vpt_amd.transfer_function import TransferFunction
    tf = TransferFunction(None)
    assert (tf.transferFunctionWidth, tf.transferFunctionHeight) == (256, 256) and tf.bumps == []
    k = tf.addBump()
    assert k == 0 and tf.bumps[0] == FIXTURE["files"]["default_bump"]["bumps"][0]          # addBump() defaults: TransferFunction.js:129-144
    tf.addBump({"position": {"x": 0.1}, "color": {"a": 0.25}})
    assert tf.bumps[1]["position"] == {"x": 0.1, "y": 0.5} and tf.bumps[1]["color"] == {"r": 1.0, "g": 0.0, "b": 0.0, "a": 0.25}
    path = tmp_path / "TransferFunction.json"
    tf.save(str(path))
    assert json.loads(path.read_text()) == tf.bumps                                          # the file IS the bump array (:82-84)
    back = TransferFunction(None).load(str(path))
    assert back.bumps == tf.bumps and (back.packed() == tf.packed()).all()
    tf.removeBump(0)
    assert len(tf.bumps) == 1
    tf.removeAllBumps()
    assert tf.bumps == [] and tf.packed().shape == (0, 8)
    with pytest.raises(ValueError):
        TransferFunction(None).loads('{"not": "an array"}')
    for name, entry in FIXTURE["files"].items():                                             # the reference's Save format loads unchanged
        assert TransferFunction(None).loads(json.dumps(entry["bumps"])).bumps == json.loads(json.dumps(entry["bumps"]), parse_int=float), name


@pytest.mark.gpu
def test_hip_rasteriser_equals_the_oracle(gpu_ctx, oracle):
    import vpt_amd
    for name, entry in FIXTURE["files"].items():
        for key, want in entry["sha256"].items():
            size, form = key.split("_")
            w, h = (int(x) for x in size.split("x"))
            tf = vpt_amd.TransferFunction(gpu_ctx, entry["bumps"], w, h)
            t = tf.texture(unpremultiply=(form == "unpremultiplied"))
            assert t.shape == (h, w, 4) and t.dtype == np.uint8
            assert hashlib.sha256(t.tobytes()).hexdigest() == want, (name, key)
    rng = np.random.default_rng(5)
    for case in range(40):
        n = int(rng.integers(0, 9))
        w, h = (int(rng.integers(1, 300)), int(rng.integers(1, 300))) if case % 4 else (1, 1)
        tf = vpt_amd.TransferFunction(gpu_ctx, None, w, h)
        for _ in range(n):
            tf.addBump({"position": {"x": rng.uniform(-0.2, 1.2), "y": rng.uniform(-0.2, 1.2)},
                        "size": {"x": float(10.0 ** rng.uniform(-3, 0.5)) * (1 if rng.uniform() < 0.9 else -1), "y": float(10.0 ** rng.uniform(-3, 0.5))},
                        "color": {"r": rng.uniform(-0.1, 1.3), "g": rng.uniform(0, 1), "b": rng.uniform(0, 1), "a": rng.uniform(0, 1.2)}})
        for un in (True, False):
            got = tf.texture(unpremultiply=un)
            want = oracle.tf_rasterize(tf.packed(), w, h, un)
            assert (got == want).all(), "case %d (%d bumps, %d x %d, unpremultiply %s): %d texel bytes differ" % (case, n, w, h, un, int((got != want).sum()))
    # non-finite bump fields draw what IEEE arithmetic makes of them, identically on both sides; a zero size is refused
    tf = vpt_amd.TransferFunction(gpu_ctx, [{"position": {"x": float("nan")}}, {"position": {"y": float("inf")}}, {"color": {"a": float("nan")}}], 33, 9)
    assert (tf.texture() == oracle.tf_rasterize(tf.packed(), 33, 9, True)).all()
    with pytest.raises(vpt_amd.VptError):
        vpt_amd.TransferFunction(gpu_ctx, [{"size": {"x": 0.0}}]).texture()


@pytest.mark.gpu
def test_a_rasterised_transfer_function_drives_a_renderer(gpu_ctx, oracle):
    """`renderer.setTransferFunction(widget.value)` (Application.js): the rasterised texels are an ordinary 256 x 256 transfer function —
    EAM with it equals the oracle's EAM with the same texels"""
    import vpt_amd
    from vpt_amd.synthetic import GoldenRatioRng
    from conftest import orbit_camera
    from test_gpu_parity import Scene, to_frame, assert_same_bits
    tf = vpt_amd.TransferFunction(gpu_ctx, FIXTURE["files"]["three_overlapping"]["bumps"])
    texels = tf.value
    assert texels.shape == (256, 256, 4)
    sc = Scene(gpu_ctx, oracle, 32, 96, 64, tf=texels, camera=orbit_camera(96 / 64, 0.4, -0.2, 1.8))
    r = sc.renderer('eam')
    r.slices = 40; r.extinction = 60
    o = oracle.OracleRenderer('eam', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    for _ in range(3):
        r.render()
        o.render(to_frame(oracle, sc, r._u))
    acc = r.read(vpt_amd._native.BUFFER_ACCUM)
    assert_same_bits(acc, o.acc.reshape(sc.h, sc.w, 4), "EAM accumulation with the rasterised transfer function")
    assert acc[:, :, :3].max() > 0                      # (the bumps' colours reach the image)
    r.destroy(); sc.gvol.destroy()
