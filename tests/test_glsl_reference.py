"""The CPU oracle (oracle/vpt_oracle.c, oracle/vpt_tonemap_oracle.c) against what the REFERENCE'S OWN SHADER TEXT computes.

tests/golden/glsl_r04.json was produced by executing the reference's programs — src/glsl/renderers/{MIP,EAM,MCS,MCM,ISO,Depth,LAO,DOS}
Renderer.glsl, src/glsl/tonemappers/*.glsl and src/glsl/TransferFunction.glsl with their mixins, read from the reference tree, cooked as
src/js/WebGL.js:85-99 cooks them —
with the GLSL interpreter of oracle/glsl_interp.py (tests/golden/make_glsl_fixtures.py; the shader text itself is not in this
repository).  So the formulas, the control flow, the order of the random draws and the meaning of every uniform are the reference's, not a
reading of them.  The oracle follows the numeric contract: explicit fma where the shader writes a * b + c, rcp_nr for
some divisions — the interpreter evaluates the shader text literally in fp32 — so the comparison is within a few ulp per pixel where
the pixel's control flow agrees, and a stochastic pixel whose comparison against a random draw falls the other way may differ: the
tests bound how many do (none, on these scenes, for most buffers).

A live part (skipped where the reference tree is absent, e.g. on the GPU box) re-executes a few fragments from the reference tree and
compares them with the committed fixture: the fixture is what the reference's text gives today."""
import base64
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "glsl_r04.json")))
GLSL_ROOT = "/root/reference/src/glsl"


def arr(b64, dtype, shape):
    return np.frombuffer(base64.b64decode(b64), dtype=dtype).reshape(shape).copy()


def scene(oracle):
    s = FX["scene"]
    vol = arr(s["volume_u8"], np.uint8, s["volume_dims_zyx"])
    tf = arr(s["tf_rgba8"], np.uint8, s["tf_shape"])
    env = arr(s["env_rgba8"], np.uint8, s["env_shape"])
    m = arr(s["mvp_inverse_f32"], np.float32, (16,))
    return oracle.OracleScene(vol, s["filter"], tf=tf, env=env), m, s["width"], s["height"]


def close(got, want, rel, abs_, what, max_outliers=0):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = ~(np.abs(got - want) <= abs_ + rel * np.abs(want))
    bad &= ~(np.isnan(got) & np.isnan(want))
    n = int(bad.reshape(bad.shape[0] * bad.shape[1], -1).any(axis=1).sum()) if bad.ndim >= 2 else int(bad.sum())
    assert n <= max_outliers, "%s: %d pixels differ (allowed %d); worst |d| = %g" % (what, n, max_outliers, float(np.nanmax(np.abs(got - want) * bad)))
    return n


def test_mip_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["mip"]
    o = oracle.OracleRenderer("mip", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    assert (o.acc.reshape(H, W) == np.rint(arr(fx["reset"]["acc"], np.float32, (H, W)) * 255)).all()
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        o.render(oracle.make_frame(W, H, m, offset=u["offset"], steps=round(1.0 / u["step"])))
        # R8 attachments: the maximum of transfer-function alphas along the ray, stored as UNORM8 — byte for byte
        close(o.frame.reshape(H, W, 1), np.rint(arr(f["frame"], np.float32, (H, W, 1)) * 255), 0, 0, "MIP frame %d" % k)
        close(o.acc.reshape(H, W, 1), np.rint(arr(f["acc"], np.float32, (H, W, 1)) * 255), 0, 0, "MIP accumulator %d" % k)
        close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 0, 0, "MIP image %d" % k)
        assert np.rint(arr(f["frame"], np.float32, (H, W)) * 255).max() > 100       # (the rays do meet the volume)


def test_eam_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["eam"]
    o = oracle.OracleRenderer("eam", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    assert (o.acc.reshape(H, W, 4) == np.rint(arr(fx["reset"]["acc"], np.float32, (H, W, 4)) * 255)).all()
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        o.render(oracle.make_frame(W, H, m, offset=u["offset"], steps=round(1.0 / u["step"]), extinction=u["extinction"], mix=u["mix"]))
        want = np.rint(arr(f["frame"], np.float32, (H, W, 4)) * 255)
        close(o.frame.reshape(H, W, 4), want, 0, 0, "EAM frame %d" % k)                   # RGBA8, byte for byte
        close(o.acc.reshape(H, W, 4), np.rint(arr(f["acc"], np.float32, (H, W, 4)) * 255), 0, 0, "EAM accumulator %d" % k)
        close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 0, 0, "EAM image %d" % k)
        assert want[:, :, :3].max() > 60 and (want[:, :, 3] == 255).all()


def test_mcs_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["mcs"]
    o = oracle.OracleRenderer("mcs", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    close(o.acc.reshape(H, W, 4), arr(fx["reset"]["acc"], np.float32, (H, W, 4)), 0, 0, "MCS reset")
    flipped = 0
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], light_dir=u["light"], mix=u["mix"]))
        want = arr(f["frame"], np.float32, (H, W, 4))
        # a pixel is a chain of random draws compared with transfer-function alphas: equal draws (the PCG stream is integer arithmetic),
        # values within rounding — unless a comparison falls the other way, which the counts below bound
        flipped += close(o.frame.reshape(H, W, 4), want, 3e-4, 2e-6, "MCS frame %d" % k, max_outliers=0)
        close(o.acc.reshape(H, W, 4), arr(f["acc"], np.float32, (H, W, 4)), 2e-4, 2e-6, "MCS accumulator %d" % k, max_outliers=flipped)
        close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 2e-3, 1e-4, "MCS image %d" % k, max_outliers=flipped)
        inside = (np.abs(want[:, :, :3] - want[0, 0, :3]).sum(axis=2) > 1e-3).sum()
        assert inside > 40, inside                                            # (pixels whose ray scattered in the volume)
    assert flipped == 0


def test_mcm_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["mcm"]
    o = oracle.OracleRenderer("mcm", sc, W, H)
    o.reset(oracle.make_frame(W, H, m, seed=FX["scene"]["mcm_reset_seed"]))
    names = ["position", "direction (+ bounces)", "transmittance", "radiance (+ samples)"]
    for q in range(4):
        # (the position of a ray that misses the cube is from + tnear * direction with tnear in the hundreds: ill-conditioned, 1e-4 relative)
        close(o.state[q].reshape(H, W, 4), arr(fx["reset"]["state"][q], np.float32, (H, W, 4)), 1e-3 if q == 0 else 2e-5, 2e-6, "MCM reset %s" % names[q])
    diverged = np.zeros((H, W), bool)
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], anisotropy=u["anisotropy"], max_bounces=u["max_bounces"], mcm_steps=u["steps"]))
        for q in range(4):
            got, want = o.state[q].reshape(H, W, 4).astype(np.float64), arr(f["state"][q], np.float32, (H, W, 4)).astype(np.float64)
            bad = (~(np.abs(got - want) <= 1e-4 + (2e-3 if q == 0 else 5e-4) * np.abs(want))).any(axis=2)
            diverged |= bad
        # the integer parts of the state — bounces so far, paths ended — are exact wherever the photon's history is the same
        same = ~diverged
        assert (o.state[1].reshape(H, W, 4)[same][:, 3] == arr(f["state"][1], np.float32, (H, W, 4))[same][:, 3]).all(), "bounces, pass %d" % k
        assert (o.state[3].reshape(H, W, 4)[same][:, 3] == arr(f["state"][3], np.float32, (H, W, 4))[same][:, 3]).all(), "samples, pass %d" % k
        assert diverged.sum() == 0, "MCM pass %d: %d of %d photons took another branch than the shader text's" % (k, diverged.sum(), W * H)
        got, want = o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4))
        assert (np.abs(got - want)[same] <= 2e-3 + 2e-3 * np.abs(want[same])).all(), "MCM image, pass %d" % k
    # the scene exercises every branch of the event loop: paths ended (samples), scattering (bounces), photons inside the volume
    last = fx["frames"][-1]["state"]
    assert arr(last[3], np.float32, (H, W, 4))[:, :, 3].max() >= 3 and arr(last[1], np.float32, (H, W, 4))[:, :, 3].max() >= 1
    pos = arr(last[0], np.float32, (H, W, 4))[:, :, :3]
    assert ((pos > 0) & (pos < 1)).all(axis=2).sum() > 20


def test_iso_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["iso"]
    o = oracle.OracleRenderer("iso", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    close(o.acc.view(np.float16).reshape(H, W, 4).astype(np.float32), arr(fx["reset"]["acc"], np.float32, (H, W, 4)), 0, 0, "ISO reset")
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        # (the oracle's frame carries uSteps twice: the loop count and the shader's 1.0 / float(uSteps), ISORenderer.glsl:64)
        o.render(oracle.make_frame(W, H, m, offset=u["offset"], steps=u["steps"], mcm_steps=u["steps"], isovalue=u["isovalue"], light_dir=u["light"], gradient_step=u["gradient_step"]))
        want = arr(f["frame"], np.float32, (H, W, 4))
        # the closest hit (position, distance) in half floats: one half ulp
        close(o.frame.view(np.float16).reshape(H, W, 4).astype(np.float32), want, 1e-3, 1e-3, "ISO closest hit of frame %d" % k, max_outliers=1)
        close(o.acc.view(np.float16).reshape(H, W, 4).astype(np.float32), arr(f["acc"], np.float32, (H, W, 4)), 1e-3, 1e-3, "ISO accumulated closest hit %d" % k, max_outliers=1)
        close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 2e-2, 4e-3, "ISO shaded image %d" % k, max_outliers=2)
        assert (want[:, :, 3] > 0).sum() > 20                                  # (rays that found the isosurface)


def test_depth_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["depth"]
    o = oracle.OracleRenderer("depth", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    close(o.acc.reshape(H, W, 1), arr(fx["reset"]["acc"], np.float32, (H, W, 1)), 0, 0, "Depth reset")
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        o.render(oracle.make_frame(W, H, m, offset=u["offset"], steps=round(1.0 / u["step"]), extinction=u["extinction"], threshold=u["threshold"], mix=u["mix"]))
        want = arr(f["frame"], np.float32, (H, W, 1))
        close(o.frame.reshape(H, W, 1), want, 1e-5, 1e-6, "Depth frame %d" % k, max_outliers=1)
        close(o.acc.reshape(H, W, 1), arr(f["acc"], np.float32, (H, W, 1)), 1e-5, 1e-6, "Depth accumulator %d" % k, max_outliers=1)
        close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 2e-3, 1e-3, "Depth image %d" % k, max_outliers=1)
        assert len(np.unique(want)) > 30


def test_lao_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["lao"]
    o = oracle.OracleRenderer("lao", sc, W, H)
    o.lao = oracle.lao_params(**fx["lao"])
    o.reset(oracle.make_frame(W, H, m))
    assert (o.acc.reshape(H, W, 4) == np.rint(arr(fx["reset"]["acc"], np.float32, (H, W, 4)) * 255)).all()
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        o.render(oracle.make_frame(W, H, m, offset=u["offset"], steps=round(1.0 / u["step"]), extinction=u["extinction"]))
        want = np.rint(arr(f["frame"], np.float32, (H, W, 4)) * 255)
        # RGBA8, byte for byte — although the shader's occlusion and shadow rays draw their directions from fract(cos(x) * 1235.68)
        close(o.frame.reshape(H, W, 4), want, 0, 0, "LAO frame %d" % k)
        close(o.acc.reshape(H, W, 4), np.rint(arr(f["acc"], np.float32, (H, W, 4)) * 255), 0, 0, "LAO accumulator %d" % k)
        close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 0, 0, "LAO image %d" % k)
        assert want[:, :, :3].max() > 40


def test_dos_follows_the_reference_shader(oracle):
    sc, m, W, H = scene(oracle)
    fx = FX["renderers"]["dos"]
    sw = fx["sweep"]
    o = oracle.OracleRenderer("dos", sc, W, H)
    fr = oracle.make_frame(W, H, m, steps=sw["steps"], extinction=sw["extinction"])
    assert np.float32(fr.step) == np.float32(sw["slice_distance"])
    o.reset(fr)
    close(o.color[o.cur].reshape(H, W, 4), arr(fx["reset"]["color"], np.float32, (H, W, 4)), 0, 0, "DOS reset colour")
    close(o.occlusion[o.cur].reshape(H, W, 1), arr(fx["reset"]["occlusion"], np.float32, (H, W, 1)), 0, 0, "DOS reset occlusion")
    lit = 0
    for k, (sl, f) in enumerate(zip(sw["slices"], fx["slices"])):
        o.integrate_slices(fr, [sl], sw["samples"])
        want = arr(f["color"], np.float32, (H, W, 4))
        close(o.color[o.cur].reshape(H, W, 4), want, 2e-4, 2e-6, "DOS colour after slice %d" % k)
        close(o.occlusion[o.cur].reshape(H, W, 1), arr(f["occlusion"], np.float32, (H, W, 1)), 2e-4, 2e-6, "DOS occlusion after slice %d" % k)
        lit = max(lit, int((want[:, :, 3] > 0).sum()))
    o.render_frame(fr)
    close(o.image_f16().astype(np.float32), arr(fx["image"], np.float32, (H, W, 4)), 2e-3, 1e-3, "DOS image")
    assert lit > 40                                                           # (pixels whose slices met the volume)


def test_transfer_function_bumps_follow_the_reference_shader(oracle):
    """the widget's fragment program (src/glsl/TransferFunction.glsl) executed per bump, blended as the GL blends: what
    vpo_tf_rasterize (and through it vpt_transfer_function_rasterize) restates — rows flipped: the fixture keeps the framebuffer's order"""
    t = FX["transfer_function"]
    W, H = t["width"], t["height"]
    bumps = np.array([[b["position"]["x"], b["position"]["y"], b["size"]["x"], b["size"]["y"], b["color"]["r"], b["color"]["g"], b["color"]["b"], b["color"]["a"]]
                      for b in t["bumps"]], np.float32)
    got = oracle.tf_rasterize(bumps, W, H, unpremultiply=False)[::-1]
    want = arr(t["canvas_rgba8"], np.uint8, (H, W, 4))
    assert (got == want).all(), int((got != want).sum())
    assert want[:, :, 3].max() > 150 and (want[:, :, 3] == 0).any()


def test_tone_mappers_follow_the_reference_shaders(oracle):
    t = FX["tonemappers"]
    img = arr(t["image_f16"], np.float16, t["image_shape"])
    p = t["params"]
    params = dict(low=p["uLow"], mid=p["uMid"], high=p["uHigh"], saturation=p["uSaturation"], min=p["uMin"], max=p["uMax"], exposure=p["uExposure"], gamma=p["uGamma"])
    worst = {}
    for name, entry in t["out"].items():
        want = arr(entry["rgba8"], np.uint8, (img.shape[0], img.shape[1], 4)).astype(np.int32)
        got = oracle.tonemap(name.lower(), img, **params).reshape(want.shape).astype(np.int32)
        worst[name] = int(np.abs(got - want).max())
        assert worst[name] == 0, (name, worst[name])                          # RGBA8, byte for byte: the operators are straight-line arithmetic
    assert len(worst) == 10


@pytest.mark.skipif(not os.path.isdir(GLSL_ROOT), reason="the reference tree is not here (GPU box): the committed fixture stands")
def test_the_fixture_is_what_the_reference_text_gives_today(oracle):
    """re-executes the reference's MCM integrate program (the headline path) for one row of pixels and its MIP generate program for another
    from the reference tree and compares with the committed fixture, bit for bit"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_glsl_fixtures as M
    from oracle import glsl_interp as G
    parts = G.read_parts(GLSL_ROOT)
    s = FX["scene"]
    W, H = s["width"], s["height"]
    vol = arr(s["volume_u8"], np.uint8, s["volume_dims_zyx"]); tf = arr(s["tf_rgba8"], np.uint8, s["tf_shape"]); env = arr(s["env_rgba8"], np.uint8, s["env_shape"])
    m = arr(s["mvp_inverse_f32"], np.float32, (16,))
    base = {"uVolume": M.volume_sampler(vol), "uTransferFunction": M.tf_sampler(tf), "uEnvironment": M.env_sampler(env), "uMvpInverseMatrix": G.mat4(m)}
    # MIP generate, frame 0, row 7
    u = FX["renderers"]["mip"]["uniforms_per_frame"][0]
    prog = G.Program(parts, "/glsl/shaders/renderers/MIP/generate")
    un = dict(base, uStepSize=np.float32(u["step"]), uOffset=np.float32(u["offset"]))
    corners = prog.varyings(un)
    row = [float(prog.fragment(un, corners, i, 7, W, H)["oColor"]) for i in range(W)]
    want = arr(FX["renderers"]["mip"]["frames"][0]["frame"], np.float32, (H, W))[7]
    assert (M.store_unorm8(row) == want).all()
    # MCM integrate, pass 0 from the reset state, row 6
    fm = FX["renderers"]["mcm"]
    state = [arr(b, np.float32, (H, W, 4)) for b in fm["reset"]["state"]]
    u = fm["uniforms_per_frame"][0]
    prog = G.Program(parts, "/glsl/shaders/renderers/MCM/integrate")
    un = dict(base, uInverseResolution=G.vec(np.float32(1.0) / np.float32(W), np.float32(1.0) / np.float32(H)), uBlur=np.float32(0.0),
              uRandSeed=np.float32(u["seed"]), uExtinction=np.float32(u["extinction"]), uAnisotropy=np.float32(u["anisotropy"]),
              uMaxBounces=G.UInt(u["max_bounces"]), uSteps=G.UInt(u["steps"]),
              uPosition=M.state_sampler(state[0]), uDirection=M.state_sampler(state[1]), uTransmittance=M.state_sampler(state[2]), uRadiance=M.state_sampler(state[3]))
    corners = prog.varyings(un)
    names = ["oPosition", "oDirection", "oTransmittance", "oRadiance"]
    for i in range(W):
        out = prog.fragment(un, corners, i, 6, W, H)
        for q, n in enumerate(names):
            got = np.array(M.comps(out[n], 4), np.float32)
            want = arr(fm["frames"][0]["state"][q], np.float32, (H, W, 4))[6, i]
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (i, n, got, want)


def test_two_channel_volume_nearest_filter_and_bounce_limit(oracle):
    """the second scene of the fixture: an RG8 volume (both channels of texture(uVolume, p).rg, the transfer function looked up in 2-D),
    the NEAREST filter, at most one bounce — MIP, EAM byte for byte; MCS within rounding; MCM the same photon histories"""
    s0, s2 = FX["scene"], FX["scene_rg8_nearest"]
    W, H = s0["width"], s0["height"]
    vol = arr(s2["volume_u8"], np.uint8, s2["volume_shape"]); tf = arr(s2["tf_rgba8"], np.uint8, s2["tf_shape"]); env = arr(s0["env_rgba8"], np.uint8, s0["env_shape"])
    m = arr(s0["mvp_inverse_f32"], np.float32, (16,))
    sc = oracle.OracleScene(vol, s2["filter"], tf=tf, env=env)
    R = FX["renderers_rg8_nearest"]
    for kind in ("mip", "eam"):
        ch = 1 if kind == "mip" else 4
        o = oracle.OracleRenderer(kind, sc, W, H)
        o.reset(oracle.make_frame(W, H, m))
        for k, (u, f) in enumerate(zip(R[kind]["uniforms_per_frame"], R[kind]["frames"])):
            kw = dict(offset=u["offset"], steps=round(1.0 / u["step"]))
            if kind == "eam":
                kw.update(extinction=u["extinction"], mix=u["mix"])
            o.render(oracle.make_frame(W, H, m, **kw))
            want = np.rint(arr(f["acc"], np.float32, (H, W, ch)) * 255)
            close(o.acc.reshape(H, W, ch), want, 0, 0, "%s accumulator %d (RG8, NEAREST)" % (kind, k))
            close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 0, 0, "%s image %d (RG8, NEAREST)" % (kind, k))
            assert want.max() > 60
    o = oracle.OracleRenderer("mcs", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    for k, (u, f) in enumerate(zip(R["mcs"]["uniforms_per_frame"], R["mcs"]["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], light_dir=u["light"], mix=u["mix"]))
        close(o.acc.reshape(H, W, 4), arr(f["acc"], np.float32, (H, W, 4)), 3e-4, 2e-6, "MCS accumulator %d (RG8, NEAREST)" % k)
    o = oracle.OracleRenderer("mcm", sc, W, H)
    o.reset(oracle.make_frame(W, H, m, seed=s0["mcm_reset_seed"]))
    for k, (u, f) in enumerate(zip(R["mcm"]["uniforms_per_frame"], R["mcm"]["frames"])):
        assert u["max_bounces"] == 1
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], anisotropy=u["anisotropy"], max_bounces=u["max_bounces"], mcm_steps=u["steps"]))
        for q in range(4):
            got, want = o.state[q].reshape(H, W, 4).astype(np.float64), arr(f["state"][q], np.float32, (H, W, 4)).astype(np.float64)
            assert (np.abs(got - want) <= 1e-4 + (2e-3 if q == 0 else 5e-4) * np.abs(want)).all(), "MCM pass %d buffer %d (RG8, NEAREST)" % (k, q)
        assert (o.state[1].reshape(H, W, 4)[..., 3] == arr(f["state"][1], np.float32, (H, W, 4))[..., 3]).all()
        assert (o.state[3].reshape(H, W, 4)[..., 3] == arr(f["state"][3], np.float32, (H, W, 4))[..., 3]).all()
    assert arr(R["mcm"]["frames"][-1]["state"][1], np.float32, (H, W, 4))[..., 3].max() == 1      # the bounce limit was reached and held


def test_camera_inside_the_volume_with_an_environment_map(oracle):
    """the third scene: the eye inside the cube (tnear < 0), wide lens, a 6 x 5 environment map, a 1 x 1 transfer function"""
    s0, s3 = FX["scene"], FX["scene_inside"]
    W, H = s0["width"], s0["height"]
    vol = arr(s0["volume_u8"], np.uint8, s0["volume_dims_zyx"]); tf = arr(s3["tf_rgba8"], np.uint8, s3["tf_shape"]); env = arr(s3["env_rgba8"], np.uint8, s3["env_shape"])
    m = arr(s3["mvp_inverse_f32"], np.float32, (16,))
    sc = oracle.OracleScene(vol, "linear", tf=tf, env=env)
    R = FX["renderers_inside"]
    for kind in ("mip", "eam", "depth"):
        ch = {"mip": 1, "eam": 4, "depth": 1}[kind]
        o = oracle.OracleRenderer(kind, sc, W, H)
        o.reset(oracle.make_frame(W, H, m))
        for k, (u, f) in enumerate(zip(R[kind]["uniforms_per_frame"], R[kind]["frames"])):
            kw = dict(offset=u["offset"], steps=round(1.0 / u["step"]))
            if kind != "mip":
                kw.update(extinction=u["extinction"], mix=u["mix"])
            if kind == "depth":
                kw.update(threshold=u["threshold"])
            o.render(oracle.make_frame(W, H, m, **kw))
            if kind == "depth":
                close(o.acc.reshape(H, W, 1), arr(f["acc"], np.float32, (H, W, 1)), 1e-5, 1e-6, "Depth accumulator %d (inside)" % k)
            else:
                close(o.acc.reshape(H, W, ch), np.rint(arr(f["acc"], np.float32, (H, W, ch)) * 255), 0, 0, "%s accumulator %d (inside)" % (kind, k))
                close(o.image_f16().astype(np.float32), arr(f["image"], np.float32, (H, W, 4)), 0, 0, "%s image %d (inside)" % (kind, k))
    o = oracle.OracleRenderer("mcs", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    for k, (u, f) in enumerate(zip(R["mcs"]["uniforms_per_frame"], R["mcs"]["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], light_dir=u["light"], mix=u["mix"]))
        # (long shadow rays from inside: a transmittance is a product of dozens of (1 - alpha) factors, 1e-4 .. 1e-3 relative by the end)
        close(o.acc.reshape(H, W, 4), arr(f["acc"], np.float32, (H, W, 4)), 1.5e-3, 2e-6, "MCS accumulator %d (inside)" % k)
    o = oracle.OracleRenderer("mcm", sc, W, H)
    o.reset(oracle.make_frame(W, H, m, seed=s0["mcm_reset_seed"]))
    close(o.state[0].reshape(H, W, 4), arr(R["mcm"]["reset"]["state"][0], np.float32, (H, W, 4)), 1e-4, 1e-5, "MCM reset position (inside: the eye itself)")
    for k, (u, f) in enumerate(zip(R["mcm"]["uniforms_per_frame"], R["mcm"]["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], anisotropy=u["anisotropy"], max_bounces=u["max_bounces"], mcm_steps=u["steps"]))
        for q in range(4):
            got, want = o.state[q].reshape(H, W, 4).astype(np.float64), arr(f["state"][q], np.float32, (H, W, 4)).astype(np.float64)
            assert (np.abs(got - want) <= 1e-4 + (2e-3 if q == 0 else 5e-4) * np.abs(want)).all(), "MCM pass %d buffer %d (inside)" % (k, q)
        assert (o.state[3].reshape(H, W, 4)[..., 3] == arr(f["state"][3], np.float32, (H, W, 4))[..., 3]).all()
    rad = arr(R["mcm"]["frames"][-1]["state"][3], np.float32, (H, W, 4))
    assert len(np.unique(np.round(rad[..., 0], 3))) > 30                      # (the environment map colours the escaping paths)


def test_parameters_at_the_ends_of_their_ranges(oracle):
    """the first scene with a step count that is no power of two (the MIP loop counts by adding 1 / steps in fp32), the EAM alpha above 1
    (its renormalisation branch), a dense medium for the trackers, |g| = 0.9, a bounce limit of 0, three / eleven events per pass"""
    sc, m, W, H = scene(oracle)
    R = FX["renderers_extremes"]
    for kind in ("mip", "eam", "depth"):
        ch = {"mip": 1, "eam": 4, "depth": 1}[kind]
        o = oracle.OracleRenderer(kind, sc, W, H)
        o.reset(oracle.make_frame(W, H, m))
        for k, (u, f) in enumerate(zip(R[kind]["uniforms_per_frame"], R[kind]["frames"])):
            fr = oracle.make_frame(W, H, m, offset=u["offset"], steps=1, **({} if kind == "mip" else dict(extinction=u["extinction"], mix=u["mix"])),
                                   **(dict(threshold=u["threshold"]) if kind == "depth" else {}))
            fr.step = float(np.float32(u["step"]))                             # the uniform itself (1 / steps as the host rounds it)
            o.render(fr)
            if kind == "depth":
                close(o.acc.reshape(H, W, 1), arr(f["acc"], np.float32, (H, W, 1)), 1e-5, 1e-6, "Depth accumulator %d (extremes)" % k)
            else:
                close(o.frame.reshape(H, W, ch), np.rint(arr(f["frame"], np.float32, (H, W, ch)) * 255), 0, 0, "%s frame %d (extremes)" % (kind, k))
                close(o.acc.reshape(H, W, ch), np.rint(arr(f["acc"], np.float32, (H, W, ch)) * 255), 0, 0, "%s accumulator %d (extremes)" % (kind, k))
    o = oracle.OracleRenderer("mcs", sc, W, H)
    o.reset(oracle.make_frame(W, H, m))
    for k, (u, f) in enumerate(zip(R["mcs"]["uniforms_per_frame"], R["mcs"]["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], light_dir=u["light"], mix=u["mix"]))
        close(o.acc.reshape(H, W, 4), arr(f["acc"], np.float32, (H, W, 4)), 1.5e-3, 2e-6, "MCS accumulator %d (extremes)" % k)
    o = oracle.OracleRenderer("mcm", sc, W, H)
    o.reset(oracle.make_frame(W, H, m, seed=FX["scene"]["mcm_reset_seed"]))
    for k, (u, f) in enumerate(zip(R["mcm"]["uniforms_per_frame"], R["mcm"]["frames"])):
        o.render(oracle.make_frame(W, H, m, seed=u["seed"], extinction=u["extinction"], anisotropy=u["anisotropy"], max_bounces=u["max_bounces"], mcm_steps=u["steps"]))
        for q in range(4):
            got, want = o.state[q].reshape(H, W, 4).astype(np.float64), arr(f["state"][q], np.float32, (H, W, 4)).astype(np.float64)
            assert (np.abs(got - want) <= 1e-4 + (2e-3 if q == 0 else 5e-4) * np.abs(want)).all(), "MCM pass %d buffer %d (extremes)" % (k, q)
        assert (o.state[1].reshape(H, W, 4)[..., 3] == arr(f["state"][1], np.float32, (H, W, 4))[..., 3]).all()
        assert (o.state[3].reshape(H, W, 4)[..., 3] == arr(f["state"][3], np.float32, (H, W, 4))[..., 3]).all()
    pos = arr(R["mcm"]["frames"][-1]["state"][0], np.float32, (H, W, 4))[..., :3]
    assert ((pos > 0) & (pos < 1)).all(axis=2).sum() > 40                      # (the dense medium holds the photons inside the volume)


# ---- the HIP library itself against the reference's text (not only through the oracle) -----------------------------------------------
@pytest.mark.gpu
def test_hip_library_against_the_reference_text(gpu_ctx, oracle):
    """MIP (byte for byte) and MCM (the same photon histories) of libvpt_hip.so through the Python host, on the fixture's scene with the
    fixture's seeds: the product path held directly to what the reference's shader text computes"""
    import vpt_amd
    from vpt_amd import _native as N
    sys_path = os.path.join(ROOT, "tests", "golden")
    import sys
    sys.path.insert(0, sys_path)
    import make_glsl_fixtures as M
    s = FX["scene"]
    W, H = s["width"], s["height"]
    vol = arr(s["volume_u8"], np.uint8, s["volume_dims_zyx"]); tf = arr(s["tf_rgba8"], np.uint8, s["tf_shape"]); env = arr(s["env_rgba8"], np.uint8, s["env_shape"])
    m = arr(s["mvp_inverse_f32"], np.float32, (16,))
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, "linear")

    class Camera:                                          # the host computes the matrix from a camera: hand it the fixture's
        pass

    def renderer(kind, seeds):
        it = iter(seeds)
        r = vpt_amd.RendererFactory(kind)(gpu_ctx, gvol, M.camera_node(W / H, 0.55, -0.3, 1.75), env, {'resolution': (W, H), 'transform': vpt_amd.Transform(vpt_amd.Node()), 'rng': lambda: next(it)})
        r.setTransferFunction(tf)
        return r
    # MIP
    fx = FX["renderers"]["mip"]
    r = renderer('mip', [u["offset"] for u in fx["uniforms_per_frame"]])
    r.steps = round(1.0 / fx["uniforms_per_frame"][0]["step"])
    r.reset()
    for k, f in enumerate(fx["frames"]):
        r.render()
        assert (np.array(list(r._u.mvp_inverse), np.float32).view(np.uint32) == m.view(np.uint32)).all()
        assert (r.read(N.BUFFER_ACCUM).reshape(H, W) == np.rint(arr(f["acc"], np.float32, (H, W)) * 255)).all(), "MIP accumulator %d" % k
        assert (r.getTexture().astype(np.float32) == arr(f["image"], np.float32, (H, W, 4))).all(), "MIP image %d" % k
    r.destroy()
    # MCM
    fx = FX["renderers"]["mcm"]
    r = renderer('mcm', [s["mcm_reset_seed"]] + [u["seed"] for u in fx["uniforms_per_frame"]])
    u0 = fx["uniforms_per_frame"][0]
    r.extinction = u0["extinction"]; r.bounces = u0["max_bounces"]; r.steps = u0["steps"]
    r.reset()
    bufs = [N.BUFFER_MCM_POSITION, N.BUFFER_MCM_DIRECTION, N.BUFFER_MCM_TRANSMITTANCE, N.BUFFER_MCM_RADIANCE]
    for k, (u, f) in enumerate(zip(fx["uniforms_per_frame"], fx["frames"])):
        r.anisotropy = u["anisotropy"]
        r.render()
        for q, b in enumerate(bufs):
            got, want = r.read(b).reshape(H, W, 4).astype(np.float64), arr(f["state"][q], np.float32, (H, W, 4)).astype(np.float64)
            assert (np.abs(got - want) <= 1e-4 + (2e-3 if q == 0 else 5e-4) * np.abs(want)).all(), "MCM pass %d buffer %d" % (k, q)
        assert (r.read(bufs[1])[..., 3] == arr(f["state"][1], np.float32, (H, W, 4))[..., 3]).all(), "bounces, pass %d" % k
        assert (r.read(bufs[3])[..., 3] == arr(f["state"][3], np.float32, (H, W, 4))[..., 3]).all(), "paths ended, pass %d" % k
    r.destroy(); gvol.destroy()
