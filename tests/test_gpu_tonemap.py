"""GPU: the ten tone mappers (vpt_tonemapper_* through the C-ABI) against the CPU oracle, bit for bit, and against the
committed golden fixture (no oracle).  SURVEY.md section 8f row 1."""
import json
import os

import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import Transform, Node, default_camera
from vpt_amd.synthetic import sphere_volume, colour_tf, GoldenRatioRng

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = ['artistic', 'range', 'reinhard', 'reinhard2', 'uncharted2', 'filmic', 'unreal', 'aces', 'lottes', 'uchimura']


def same(got, want, what):
    g, w = np.asarray(got).reshape(-1), np.asarray(want).reshape(-1)
    assert g.shape == w.shape, what
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, "%s: %d of %d bytes differ, first at %d: got %r want %r" % (what, bad.size, g.size, bad[0], g[bad[0]], w[bad[0]])


def test_exp_pow_probes_bit_exact(gpu_ctx, oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    x = np.concatenate([np.linspace(-110, 95, 40001), rng.uniform(-3, 3, 20000),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 89.0, 89.00001, -104.0, -104.00001, -87.4, -100.0, 88.7]]).astype(np.float32)
    got = gpu_ctx.probe_math(N.PROBE_EXP, x)
    want = np.array([L.vpo_expf(float(v)) for v in x], dtype=np.float32)
    assert (got.view(np.uint32)[~np.isnan(want)] == want.view(np.uint32)[~np.isnan(want)]).all()
    assert np.isnan(got[np.isnan(want)]).all()
    base = np.concatenate([rng.uniform(0, 70, 30000), rng.uniform(0, 1, 10000) ** 4,
                           [0.0, -0.0, 1.0, -1.0, np.inf, np.nan, 6e-8, 65504.0, 1e-30, 1e30]]).astype(np.float32)
    expo = np.concatenate([rng.uniform(0.05, 3.0, 40000), [0.4545, 2.2, 0.0, 1.0, 0.5, 1.0, 0.4545, 2.2, 1.6, 1.33]]).astype(np.float32)
    pairs = np.stack([base, expo], -1).reshape(-1)
    got = gpu_ctx.probe_math(N.PROBE_POW, pairs)
    want = np.array([L.vpo_powf(float(a), float(b)) for a, b in zip(base, expo)], dtype=np.float32)
    nan = np.isnan(want)
    assert np.isnan(got[nan]).all()
    assert (got.view(np.uint32)[~nan] == want.view(np.uint32)[~nan]).all()


def hdr_image(w, h, seed=3):
    rng = np.random.default_rng(seed)
    img = np.ones((h, w, 4), dtype=np.float32)
    img[..., :3] = rng.uniform(0, 5.0, size=(h, w, 3)) ** 3 / 12.0
    img[..., 3] = np.where(rng.uniform(size=(h, w)) < 0.9, 1.0, rng.uniform(0, 2, size=(h, w)))
    flat = img.reshape(-1, 4)
    specials = np.array([0.0, -0.0, -0.5, np.inf, -np.inf, np.nan, 6e-8, 6.1e-5, 65504.0, 0.004, 0.22, 0.532, 0.62, 1.0], dtype=np.float32)
    idx = rng.choice(flat.shape[0], size=3 * specials.size, replace=False)
    for n, i in enumerate(idx):
        flat[i, n % 3] = specials[n // 3]
    return img.astype(np.float16)


PARAMS = {
    'artistic': [{}, {'low': 0.1, 'mid': 0.3, 'high': 2.5, 'saturation': 0.4, 'gamma': 1.8}, {'low': 1.0, 'high': 1.0},
                 {'saturation': 1, 'low': 0.1, 'mid': 0.7, 'high': 3.0}, {'saturation': 1, 'low': 0.2, 'high': 0.2},
                 {'saturation': 1, 'low': 0.0, 'high': 1e-30, 'mid': 5e-31}],
    'range': [{}, {'min': -0.5, 'max': 3.0, 'gamma': 1.0}, {'gamma': 0.0}],
}


@pytest.mark.parametrize("kind", KINDS)
def test_tonemapper_image_parity(gpu_ctx, oracle, kind):
    """every mapper on an HDR image with zeros, negatives, inf, NaN and subnormal halfs: bit-identical to the oracle"""
    w, h = 150, 70
    img = hdr_image(w, h)
    T = vpt_amd.ToneMapperFactory(kind)
    tm = T(gpu_ctx, img, {'resolution': (w, h)})
    for params in PARAMS.get(kind, [{}, {'exposure': 2.5, 'gamma': 1.7}, {'exposure': 0.0, 'gamma': 2.2}]):
        for k, v in params.items():
            setattr(tm, k, v)
        full = {p['name']: getattr(tm, p['name']) for p in tm.properties}
        want = oracle.tonemap(kind, img, **full)
        # direct evaluation per pixel, then the 65 536-entry byte-table form (a no-op for Artistic); twice: the cached table
        for mode in (N.TONEMAPPER_TABLE_NEVER, N.TONEMAPPER_TABLE_ALWAYS, N.TONEMAPPER_TABLE_ALWAYS):
            tm.set_option(N.TONEMAPPER_OPTION_TABLE, mode)
            tm.render()
            got = tm.getTexture()
            assert got.shape == (h, w, 4) and got.dtype == np.uint8
            same(got, want, "%s %r table mode %d" % (kind, params, mode))
    tm.destroy()


def test_golden_fixture_without_oracle(gpu_ctx):
    fx = json.load(open(os.path.join(HERE, "golden", "tonemap_r01.json")))
    src = np.array(fx["source_rgba16f_bits"], dtype=np.uint16).view(np.float16).reshape(1, -1, 4)
    for case in fx["cases"]:
        tm = vpt_amd.ToneMapperFactory(case["kind"])(gpu_ctx, src, {'resolution': (src.shape[1], 1)})
        for k, v in case["params"].items():
            setattr(tm, k, v)
        tm.render()
        same(tm.getTexture(), np.array(case["rgba8"], dtype=np.uint8), "fixture %s %r" % (case["kind"], case["params"]))
        tm.destroy()


def test_properties_mirror_the_reference(gpu_ctx):
    a = vpt_amd.ArtisticToneMapper(gpu_ctx, None, {'resolution': 8})
    assert [(p['name'], p['value']) for p in a.properties] == [('low', 0), ('high', 1), ('mid', 0.5), ('saturation', 1), ('gamma', 2.2)]   # ArtisticToneMapper.js:15-49
    assert a.properties[2]['min'] == 0.00001 and a.properties[2]['max'] == 0.99999 and a.properties[2]['type'] == 'slider'
    r = vpt_amd.RangeToneMapper(gpu_ctx, None, {'resolution': 8})
    assert [(p['name'], p['value']) for p in r.properties] == [('min', 0), ('max', 1), ('gamma', 2.2)]                                     # RangeToneMapper.js:14-34
    for kind in KINDS[2:]:
        t = vpt_amd.ToneMapperFactory(kind)(gpu_ctx, None)
        assert [(p['name'], p['value'], p.get('min')) for p in t.properties] == [('exposure', 1, 0), ('gamma', 2.2, 0)]                    # ReinhardToneMapper.js:14-29
        assert t._size() == (512, 512)                                                                                                    # AbstractToneMapper.js:15
        t.destroy()
    # the placeholder texture of RenderingContext.js:176-181 (1x1 white): every texel maps white
    a.render()
    out = a.getTexture()
    assert out.shape == (8, 8, 4) and (out == 255).all()
    a.destroy(); r.destroy()


@pytest.mark.parametrize("rkind,tkind", [("mcm", "artistic"), ("mcs", "reinhard"), ("eam", "range"), ("mip", "aces")])
def test_renderer_to_tonemapper_chain(gpu_ctx, oracle, rkind, tkind):
    """RenderingContext.render(): renderer.render(); toneMapper.render() — the tone mapper reads the render buffer in HBM"""
    w, h = 112, 80
    vol = sphere_volume(32, noise=40.0)
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')
    cam = default_camera(w / h)
    r = vpt_amd.RendererFactory(rkind)(gpu_ctx, gvol, cam, None, {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    r.setTransferFunction(colour_tf(64, 1))
    if rkind in ('mcs', 'mcm'):
        r.extinction = 8
    r.reset()
    tm = vpt_amd.ToneMapperFactory(tkind)(gpu_ctx, r, {'resolution': (w, h)})     # chooseToneMapper: texture = renderer.getTexture()
    for _ in range(3):
        r.render(); tm.render()
    same(tm.getTexture(), oracle.tonemap(tkind, r.getTexture(), **{p['name']: p['value'] for p in tm.properties}), "%s -> %s" % (rkind, tkind))
    # setResolution on both, as RenderingContext.js:219-228 does
    r.setResolution((64, 48)); tm.setResolution((64, 48)); tm.setTexture(r)
    r.render(); tm.render()
    assert tm.getTexture().shape == (48, 64, 4)
    same(tm.getTexture(), oracle.tonemap(tkind, r.getTexture()), "after setResolution")
    # a mismatch is refused loudly rather than resampled
    tm.setResolution((32, 32))
    with pytest.raises(vpt_amd.VptError, match="resampling"):
        tm.render()
    tm.destroy(); r.destroy(); gvol.destroy()


def test_sharded_source_rows(gpu_ctx, oracle):
    """a sharded renderer's tone mapper maps the local rows; together the ranks' rows are the unsharded image"""
    w, h = 96, 72
    gvol = vpt_amd.Volume.from_array(gpu_ctx, sphere_volume(24, noise=30.0), 'linear')
    cam = default_camera(w / h)

    def run(**opts):
        o = {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()}
        o.update(opts)
        r = vpt_amd.MCMRenderer(gpu_ctx, gvol, cam, None, o)
        r.extinction = 6
        r.reset()
        tm = vpt_amd.Reinhard2ToneMapper(gpu_ctx, r, {'resolution': (w, h)})
        for _ in range(2):
            r.render()
        tm.render()
        out = (tm.getTexture(), r.global_rows())
        tm.destroy(); r.destroy()
        return out

    whole, _ = run()
    got = np.zeros_like(whole)
    for rank in range(3):
        img, rows = run(shard=(rank, 3, 8))
        got[rows[rows >= 0]] = img[rows >= 0]
    same(got, whole, "3-way sharded tone map")
    gvol.destroy()


def test_full_size_tonemap(gpu_ctx, oracle):
    """1920x1080: every mapper over a full frame equals the oracle on a band of rows; alpha stays 255"""
    w, h = 1920, 1080
    img = hdr_image(w, h, seed=9)
    for kind in KINDS:
        tm = vpt_amd.ToneMapperFactory(kind)(gpu_ctx, img, {'resolution': (w, h)})
        tm.render()                                            # AUTO: the table form at this size (except Artistic)
        out = tm.getTexture()
        same(out[500:516], oracle.tonemap(kind, img[500:516]), "%s rows 500..516" % kind)
        tm.set_option(N.TONEMAPPER_OPTION_TABLE, N.TONEMAPPER_TABLE_NEVER)
        tm.render()
        same(tm.getTexture(), out, "%s direct vs table over the whole frame" % kind)
        if kind != 'range':
            assert (out[..., 3] == 255).all()
        tm.destroy()


def test_large_frame_lds_table_form(gpu_ctx, oracle):
    """3328x2048 (6.8 Mpixel): from 6 Mpixel on the byte table is applied from LDS; whole frame == the direct form, a band
    == the oracle, for the table-with-alpha mapper (range) and two of the others"""
    w, h = 3328, 2048
    img = hdr_image(w, h, seed=21)
    for kind in ('range', 'reinhard', 'uchimura'):
        tm = vpt_amd.ToneMapperFactory(kind)(gpu_ctx, img, {'resolution': (w, h)})
        tm.render()
        out = tm.getTexture()
        same(out[1000:1008], oracle.tonemap(kind, img[1000:1008]), "%s rows 1000..1008" % kind)
        same(out[-2:], oracle.tonemap(kind, img[-2:]), "%s last rows" % kind)
        tm.set_option(N.TONEMAPPER_OPTION_TABLE, N.TONEMAPPER_TABLE_NEVER)
        tm.render()
        same(tm.getTexture(), out, "%s direct vs LDS table over the whole frame" % kind)
        tm.destroy()


def test_rendering_context_sequence(gpu_ctx, oracle):
    """the caller's sequence (RenderingContext.js:123-133,152-210,216-229): setVolume -> chooseRenderer -> chooseToneMapper ->
    N x render(), resolution change, renderer / tone mapper swaps, filter change — the frame equals the oracle chain"""
    from vpt_amd.readers import RAWReader
    vol = sphere_volume(24, noise=30.0)
    rc = vpt_amd.RenderingContext({'resolution': (88, 64), 'rng': GoldenRatioRng()})
    seen = []
    rc.addEventListener('progress', lambda e: seen.append(e.detail))
    rc.resize(88, 64)
    rc.setVolume(RAWReader(vol, {'width': 24, 'height': 24, 'depth': 24}))
    assert seen and seen[-1] == 1
    rc.render()                                                   # no renderer / tone mapper yet: a no-op (:191-193)
    rc.chooseToneMapper('artistic')                               # before any renderer: the white placeholder
    rc.chooseRenderer('mcm')
    rc.renderer.extinction = 7
    rc.renderer.reset()
    for _ in range(3):
        rc.render()
    frame = rc.getFrame()
    assert frame.shape == (64, 88, 4)
    same(frame, oracle.tonemap('artistic', rc.renderer.getTexture()), "context frame")
    assert (frame[..., 3] == 255).all() and (frame[..., :3] > 0).any()
    rc.resolution = (48, 40); rc.resize(48, 40)
    rc.render()
    assert rc.getFrame().shape == (40, 48, 4)
    rc.chooseToneMapper('aces'); rc.chooseRenderer('eam'); rc.setFilter('nearest')
    rc.camera.transform.localTranslation = [0.3, 0.2, 1.8]        # 'change' -> renderer.reset() (:42-46)
    rc.render(); rc.render()
    same(rc.getFrame(), oracle.tonemap('aces', rc.renderer.getTexture()), "after swaps")
    rc.destroy()


def test_record_animation_to_image_sequence(gpu_ctx, oracle, tmp_path):
    """RenderingContext.recordAnimationToImageSequence (RenderingContext.js:259-305), headless: CircleAnimator steps, a reset
    and `passes` frames per image, PNG files; frame i equals the same sequence driven by hand"""
    from vpt_amd.png import decode_png
    from vpt_amd.readers import RAWReader
    vol = sphere_volume(24, noise=30.0)

    def context():
        rc = vpt_amd.RenderingContext({'resolution': (72, 56), 'rng': GoldenRatioRng()})
        rc.setVolume(RAWReader(vol, {'width': 24, 'height': 24, 'depth': 24}))
        rc.chooseRenderer('eam'); rc.chooseToneMapper('reinhard')
        rc.cameraAnimator = vpt_amd.CircleAnimator(rc.camera, {'center': [0, 0, 2], 'direction': [0, 0, 1], 'radius': 0.3, 'frequency': 1})
        return rc

    rc = context()
    seen = []
    rc.addEventListener('animationprogress', lambda e: seen.append(e.detail))
    files = rc.recordAnimationToImageSequence({'directory': str(tmp_path / 'anim'), 'startTime': 0.0, 'endTime': 0.4, 'fps': 10, 'passes': 3})
    assert [os.path.basename(f) for f in files] == ['frame0000.png', 'frame0001.png', 'frame0002.png', 'frame0003.png'] and seen[-1] == 1
    rc.destroy()
    # by hand: the rng keeps running across images exactly as in the recorder
    rc = context()
    for i, f in enumerate(files):
        rc.cameraAnimator.update(0.0 + i * (1 / 10))
        rc.renderer.reset()
        for _ in range(3):
            rc.render()
        want = rc.getFrame()
        got = decode_png(open(f, 'rb').read())
        same(got, want[::-1], "animation frame %d" % i)
        assert (want[..., :3] > 0).any()
    rc.destroy()


def test_orbit_camera_turntable_through_rendering_context(gpu_ctx, oracle):
    """the context's default animator is an OrbitCameraAnimator (RenderingContext.js:54); a scripted drag / wheel / WASD
    sequence moves the camera, every move resets the renderer (:42-46), and each frame equals the oracle's for that pose"""
    from vpt_amd.readers import RAWReader
    from vpt_amd.scene import mvp_inverse_matrix
    vol = sphere_volume(24, noise=30.0)
    rc = vpt_amd.RenderingContext({'resolution': (80, 60), 'rng': GoldenRatioRng()})
    assert isinstance(rc.cameraAnimator, vpt_amd.OrbitCameraAnimator) and rc.cameraAnimator._focusDistance == 2.0
    rc.setVolume(RAWReader(vol, {'width': 24, 'height': 24, 'depth': 24}))
    rc.chooseRenderer('mip'); rc.chooseToneMapper('range')
    rc.renderer.steps = 30
    clock = [0.0]
    orbit = rc.cameraAnimator
    orbit.now = lambda: clock[0]; orbit._time = 0.0
    osc = oracle.OracleScene(vol, 'linear')
    script = [lambda: (orbit._handlePointerDown({'button': 0}), orbit._handlePointerMove({'movementX': 60, 'movementY': -25})),
              lambda: orbit._handlePointerMove({'movementX': -140, 'movementY': 80}),
              lambda: (orbit._handlePointerUp(), orbit._handleWheel({'deltaY': -300})),
              lambda: (orbit._handleKeyDown({'key': 'a'}), clock.__setitem__(0, clock[0] + 120), orbit._update(), orbit._handleKeyUp({'key': 'a'}))]
    poses = set()
    for k, act in enumerate(script):
        rc.render()
        before = rc.renderer.read(N.BUFFER_ACCUM).copy()
        act()                                                       # camera 'change' -> renderer.reset()
        assert (rc.renderer.read(N.BUFFER_ACCUM) == 0).all() and before.any()
        rng_probe = GoldenRatioRng()
        rc.renderer.rng = GoldenRatioRng()                          # restart the jitter sequence: the oracle below starts from draw 1
        o = oracle.OracleRenderer('mip', osc, 80, 60)
        m = mvp_inverse_matrix(rc.camera, rc.volumeTransform)
        o.reset(oracle.make_frame(80, 60, m))
        for _ in range(2):
            rc.render()
            o.render(oracle.make_frame(80, 60, m, steps=30, offset=np.float32(rng_probe())))
        same(rc.renderer.getTexture().view(np.uint16), o.out.reshape(60, 80, 4), "orbit pose %d" % k)
        same(rc.getFrame(), oracle.tonemap('range', rc.renderer.getTexture()), "orbit pose %d tone mapped" % k)
        poses.add(tuple(rc.camera.transform.localTranslation.tolist()))
    assert len(poses) == 4
    rc.destroy()


def test_tonemapper_survives_its_renderer(gpu_ctx, oracle):
    """destroying the bound renderer unbinds it: the tone mapper falls back to the white placeholder instead of reading freed memory"""
    gvol = vpt_amd.Volume.from_array(gpu_ctx, sphere_volume(16, noise=20.0), 'linear')
    r = vpt_amd.MIPRenderer(gpu_ctx, gvol, default_camera(1.0), None, {'resolution': 48, 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    r.reset(); r.render()
    tm = vpt_amd.RangeToneMapper(gpu_ctx, r, {'resolution': 48})
    tm.render()
    assert (tm.getTexture()[..., 0] < 255).any()
    r.destroy()
    tm.render()
    assert (tm.getTexture() == 255).all()
    tm.destroy(); gvol.destroy()


def test_renderer_survives_its_volume(gpu_ctx, oracle):
    """destroying a bound volume through the C ABI unbinds it: the next pass reports 'no ready volume' (Volume.js:107-113)"""
    import ctypes as C
    gvol = vpt_amd.Volume.from_array(gpu_ctx, sphere_volume(16, noise=20.0), 'linear')
    r = vpt_amd.MIPRenderer(gpu_ctx, gvol, default_camera(1.0), None, {'resolution': 32, 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    r.reset(); r.render()
    u = r._prepare_frame_uniforms()
    N.check(N.lib().vpt_volume_destroy(gvol.texture))          # behind the host mirror's back
    gvol.texture = None; gvol.ready = False
    assert N.lib().vpt_renderer_render(r._h, C.byref(u)) == -3  # VPT_ERR_NO_VOLUME
    assert b"no ready volume" in N.lib().vpt_last_error()
    r.destroy()


@pytest.mark.parametrize("seed", range(12))
def test_tonemapper_random_parameters(gpu_ctx, oracle, seed):
    """all ten mappers under random (also degenerate) parameters, direct and table form, on an image with specials"""
    rng = np.random.default_rng(500 + seed)
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 60))
    img = hdr_image(w, h, seed=100 + seed)

    def pick(lo, hi):
        return float(rng.choice([rng.uniform(lo, hi), 0.0, 1.0, -rng.uniform(0, 1), 1e-6, 1e6]))
    for kind in KINDS:
        tm = vpt_amd.ToneMapperFactory(kind)(gpu_ctx, img, {'resolution': (w, h)})
        for p in tm.properties:
            setattr(tm, p['name'], pick(0.05, 4.0))
        full = {p['name']: getattr(tm, p['name']) for p in tm.properties}
        want = oracle.tonemap(kind, img, **full)
        for mode in (N.TONEMAPPER_TABLE_NEVER, N.TONEMAPPER_TABLE_ALWAYS):
            tm.set_option(N.TONEMAPPER_OPTION_TABLE, mode)
            tm.render()
            same(tm.getTexture(), want, "%s %r mode %d" % (kind, full, mode))
        tm.destroy()


# ---------------------------------------------------------------------------------------------------------------------------
# VPT_TONEMAPPER_OPTION_FUSE: once the tone mapper has run in its table form on a bound renderer, the renderer's fused passes write
# the tone-mapped texel next to every RGBA16F texel they store and the tone mapper's later render() calls launch nothing
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rkind", ["mcm", "eam", "mip", "mcs", "iso", "depth", "lao"])
@pytest.mark.parametrize("tkind", ["artistic", "range", "reinhard"])
def test_fused_tone_mapping_equals_the_separate_pass(gpu_ctx, oracle, rkind, tkind):
    from vpt_amd import _native as N
    w, h = 208, 144
    vol = sphere_volume(24, noise=40.0)
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')
    from conftest import orbit_camera

    def run(fuse):
        cam = orbit_camera(w / h, 0.7, -0.3, 3.0)
        r = vpt_amd.RendererFactory(rkind)(gpu_ctx, gvol, cam, None, {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
        r.setTransferFunction(colour_tf(64, 1))
        if rkind in ('mcs', 'mcm'):
            r.extinction = 6
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.reset()
        tm = vpt_amd.ToneMapperFactory(tkind)(gpu_ctx, r, {'resolution': (w, h)})
        tm.set_option(N.TONEMAPPER_OPTION_TABLE, N.TONEMAPPER_TABLE_ALWAYS)         # (small image: AUTO would take the direct form)
        tm.set_option(N.TONEMAPPER_OPTION_FUSE, fuse)
        outs = []

        def shot(what):
            tm.render()
            got = tm.getTexture().copy()
            same(got, oracle.tonemap(tkind, r.getTexture(), **{p['name']: getattr(tm, p['name']) for p in tm.properties}), "%s -> %s fuse=%d: %s" % (rkind, tkind, fuse, what))
            outs.append(got)
        for k in range(3):
            r.render(); shot("frame %d" % k)
        r.render(); r.render(); shot("two frames without a tone-map call in between")
        name = {'artistic': 'high', 'range': 'max', 'reinhard': 'exposure'}[tkind]
        setattr(tm, name, getattr(tm, name) * 0.7)                                  # new parameters: the table is rebuilt
        shot("new parameters, same frame")
        r.render(); shot("new parameters, next frame")
        r.fused = False
        r.render(); shot("a hook-by-hook frame")                                   # the hook kernels do not tone-map
        r.fused = True
        r.render(); shot("fused again")
        r.reset(); r.render(); shot("after a reset")
        if rkind != "lao":
            r.play(3, use_graph=False); shot("an eager sequence")
        if rkind == "mcm":
            r.play(3, fused=True); shot("a fused-pass sequence")
            r.render(); shot("and a frame after it")
        tm.destroy(); r.destroy()
        return outs

    a, b = run(0), run(1)
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert (x == y).all(), (rkind, tkind, k)
    gvol.destroy()


def test_fused_tone_mapping_full_hd_mcm_tile_classes(gpu_ctx, oracle):
    """the displayed-frame path of RenderingContext.render() at the headline size: MCM (fast-math, tile classes on two streams) + the default
    Artistic tone mapper; AUTO takes the table form at this size, so the fusion is what runs"""
    from vpt_amd import _native as N
    w, h = 1920, 1080
    vol = sphere_volume(128, noise=48.0)
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')
    r = vpt_amd.MCMRenderer(gpu_ctx, gvol, default_camera(w / h), None, {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    r.set_option(N.OPTION_FAST_MATH, 1); r.set_option(N.OPTION_SPLIT_STREAMS, 2)
    r.reset()
    tm = vpt_amd.ToneMapperFactory('artistic')(gpu_ctx, r, {'resolution': (w, h)})
    for k in range(6):
        r.render(); tm.render()
        if k in (0, 1, 5):
            same(tm.getTexture(), oracle.tonemap('artistic', r.getTexture(), **{p['name']: p['value'] for p in tm.properties}), "frame %d" % k)
    tm.destroy(); r.destroy(); gvol.destroy()


@pytest.mark.parametrize("rkind,fast", [("mcm", 0), ("mcm", 1), ("eam", 0)])
@pytest.mark.parametrize("tkind", ["artistic", "range", "aces"])
def test_play_into_display_equals_the_tone_mapped_frames(gpu_ctx, oracle, rkind, fast, tkind):
    """vpt_renderer_play_into_display: a bucket of frames as the armed tone mapper shows them (RGBA8, half the bytes a collective has to move).
    MCM with the tile classes in force runs the bucket kernels with the tone mapper's table in their frame store; a moved camera (classes
    void) and the other renderers go frame by frame through the fused pass + a copy of the tone mapper's output.  Every frame must equal
    render() + toneMapper.render() of the same frame.  The renderer's frame ring stands in for the caller's bucket memory."""
    import ctypes as C
    from vpt_amd import _native as N
    from conftest import orbit_camera
    w, h = 208, 144
    vol = sphere_volume(24, noise=40.0)
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')

    def make():
        cam = orbit_camera(w / h, 0.7, -0.3, 3.0)
        r = vpt_amd.RendererFactory(rkind)(gpu_ctx, gvol, cam, None, {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
        r.setTransferFunction(colour_tf(64, 1))
        if rkind == 'mcm':
            r.extinction = 6; r.steps = 4
            r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.reset()
        tm = vpt_amd.ToneMapperFactory(tkind)(gpu_ctx, r, {'resolution': (w, h)})
        tm.set_option(N.TONEMAPPER_OPTION_TABLE, N.TONEMAPPER_TABLE_ALWAYS)
        return r, tm, cam

    # the reference sequence: render() + toneMapper.render(), frame by frame
    r, tm, cam = make()
    want = []
    r.render(); tm.render()
    for k in range(7 + 3):
        if k == 7:
            cam.transform.localTranslation = [0.3, 0.2, 1.6]; cam.transform.localRotation = [0, 0, 0, 1]
        r.render(); tm.render()
        want.append(tm.getTexture().copy())
    tm.destroy(); r.destroy()

    r, tm, cam = make()
    r.render(); tm.render()                                              # arms the tone mapper on the renderer
    # bucket memory: the frame ring of an idle MCM renderer on the same context stands in for the caller's buffers (16 slots of w*h*8 bytes)
    aux = vpt_amd.MCMRenderer(gpu_ctx, gvol, orbit_camera(w / h), None, {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    aux.reset(); aux.play(16, frames=True)
    p_, n_ = C.c_void_p(), C.c_size_t()
    N.check(N.lib().vpt_renderer_frame_ring_device(aux._h, C.byref(p_), C.byref(n_)))
    ptr, stride = p_.value, w * h * 4
    assert n_.value == 2 * stride

    def display_frames(count):
        slots = [aux.read_frame_slot(k).view(np.uint8).reshape(2, h, w, 4) for k in range((count + 1) // 2)]
        return np.concatenate(slots)[:count].copy()

    launches0 = r.bucket_launches() if rkind == 'mcm' else 0
    r.play_into_display(tm, 7, ptr, stride)
    got = display_frames(7)
    if rkind == 'mcm':
        assert r.bucket_launches() == launches0 + 1
    cam.transform.localTranslation = [0.3, 0.2, 1.6]; cam.transform.localRotation = [0, 0, 0, 1]     # no reset: frame by frame from here on
    r.play_into_display(tm, 3, ptr, stride)
    got2 = display_frames(3)
    if rkind == 'mcm':
        assert r.bucket_launches() == launches0 + 1
    for k in range(7):
        same(got[k], want[k], "%s -> %s: display frame %d of the bucket" % (rkind, tkind, k))
    for k in range(3):
        same(got2[k], want[7 + k], "%s -> %s: display frame %d behind the moved camera" % (rkind, tkind, k))
    aux.destroy(); tm.destroy(); r.destroy(); gvol.destroy()
