"""GPU: tile classes (VPT_OPTION_TILE_CLASSES, vpt_kernels.h "Tile classes") — the MCM passes of tiles none of whose camera rays
can meet the cube run k_mcm_miss on 32 of the 56 state bytes per pixel; HIT tiles run k_mcm_integrate from a tile list.

Everything a caller can read must be identical, bit for bit, with the option on and off, and identical to the CPU oracle
(contract arithmetic): all four photon-state buffers (position / transmittance of MISS tiles are materialised on demand), the
render buffer, the sample count — through reads in the middle of a sequence, a matrix that changes without a reset, a blur,
an option toggled between passes, shards, frame sequences in every mode.  VPT_OPTION_VERIFY_TILE_CLASSES counts events that
contradict the classification: 0."""
import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.synthetic import colour_tf, ramp_tf, sphere_volume, GoldenRatioRng

from conftest import orbit_camera
from test_gpu_parity import Scene, to_frame, assert_same_bits, MCM_BUFFERS, env_map

pytestmark = pytest.mark.gpu


def far_scene(gpu_ctx, oracle, w=208, h=144, env=None, cam=(0.7, -0.3, 3.2), n=24, filt="linear"):
    """a camera far enough that most 16x16 tiles miss the cube"""
    return Scene(gpu_ctx, oracle, n, w, h, filt, tf=colour_tf(48, 1), env=env, camera=orbit_camera(w / h, *cam))


def mcm_with_form(sc, form, **kw):
    """an MCM renderer whose HIT-tile kernel form is forced (VPT_HIT_KERNEL_FORM is read when the renderer is created): 1 = k_mcm_integrate,
    2 = k_mcm_integrate_early, 0 = the library's choice by the number of HIT tiles"""
    import os
    old = os.environ.get("VPT_HIT_KERNEL_FORM")
    if form:
        os.environ["VPT_HIT_KERNEL_FORM"] = str(form)
    else:
        os.environ.pop("VPT_HIT_KERNEL_FORM", None)
    try:
        return sc.renderer('mcm', **kw)
    finally:
        if old is None:
            os.environ.pop("VPT_HIT_KERNEL_FORM", None)
        else:
            os.environ["VPT_HIT_KERNEL_FORM"] = old


def all_buffers(r):
    return [r.read(b).copy() for b in MCM_BUFFERS] + [r.getTexture().copy()]


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("split,form", [(1, 0), (3, 1), (2, 2)])
def test_classes_on_and_off_give_identical_buffers(gpu_ctx, oracle, fast, split, form):
    """form: VPT_HIT_KERNEL_FORM in the environment — 1 = k_mcm_integrate on the HIT tiles, 2 = k_mcm_integrate_early (path end of the out-of-cube
    lanes under the sample's loads), 0 = chosen by the number of HIT tiles; one stream runs the general kernel on every tile"""
    sc = far_scene(gpu_ctx, oracle, env=env_map(16, 8))

    def run(classes):
        r = mcm_with_form(sc, form)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
        r.extinction = 4; r.steps = 5; r.anisotropy = 0.3
        r.reset()
        hit, miss, _ = r.tile_classes()
        if classes:
            assert miss > hit > 0, (hit, miss)
        outs = all_buffers(r)                               # the reset's buffers
        for k in range(6):
            r.render()
            if k in (0, 3):
                outs += all_buffers(r)                      # position / transmittance of MISS tiles materialised mid-sequence
        r.render(); r.render()
        outs += all_buffers(r)
        assert r.sample_count() == sc.w * sc.h * 5 * 8
        assert r.tile_classes()[2] == 0                     # no event of a MISS tile was inside the cube
        r.destroy()
        return outs

    a, b = run(0), run(1)
    for k, (x, y) in enumerate(zip(a, b)):
        assert_same_bits(y, x, "tile classes on vs off, output %d" % k)
    sc.gvol.destroy()


@pytest.mark.parametrize("env", [None, (16, 8)])
@pytest.mark.parametrize("cam", [(0.7, -0.3, 3.2), (2.4, 0.5, 2.1)])
def test_classes_against_the_oracle(gpu_ctx, oracle, env, cam):
    sc = far_scene(gpu_ctx, oracle, env=env_map(*env) if env else None, cam=cam)
    r = sc.renderer('mcm')
    r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
    r.extinction = 5; r.steps = 6
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset()
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    assert r.tile_classes()[1] > 0
    for k in range(5):
        r.render()
        o.render(to_frame(oracle, sc, r._u))
        if k in (1, 4):
            for b, s in zip(MCM_BUFFERS, o.state):
                assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "state buffer %d after pass %d" % (b, k))
            assert_same_bits(r.getTexture().view(np.uint16), o.image_f16().view(np.uint16), "render buffer")
    assert r.tile_classes()[2] == 0
    r.destroy(); sc.gvol.destroy()


class BlurredMCM(vpt_amd.MCMRenderer):
    """uBlur is always 0 in the reference (MCMRenderer.js:93,157); the boundary takes any value"""
    blur_value = 0.0

    def _prepare_integrate(self):
        u = super()._prepare_integrate()
        u.blur = float(np.float32(self.blur_value))
        return u


BlurredMCM._BASE = BlurredMCM


@pytest.mark.parametrize("fast", [0, 1])
def test_matrix_or_blur_changing_without_a_reset_voids_the_classes(gpu_ctx, oracle, fast):
    """passes with another matrix (the camera moved, nobody called reset()) or a blur: the MISS tiles' photons may enter the cube
    now — the library must notice by itself, bring the lazily kept arrays up to date and go back to the general kernel"""
    def run(classes):
        sc = far_scene(gpu_ctx, oracle)
        o = {'resolution': (sc.w, sc.h), 'transform': sc.transform, 'rng': GoldenRatioRng()}
        r = BlurredMCM(gpu_ctx, sc.gvol, sc.camera, None, o)
        r.setTransferFunction(sc.tf)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.extinction = 4; r.steps = 4
        r.reset()
        outs = []
        for _ in range(3):
            r.render()
        # the camera swings round towards the volume: former MISS tiles now look at it
        sc.camera.transform.localTranslation = [0.2, 0.1, 1.2]
        sc.camera.transform.localRotation = [0, 0, 0, 1]
        for _ in range(3):
            r.render()
        if classes:
            assert r.tile_classes()[1] == 0             # void until the next reset
        outs += all_buffers(r)
        r.reset()
        if classes:
            assert r.tile_classes()[1] >= 0
        r.render(); r.render()
        r.blur_value = 0.4                              # a blurred pass: near-plane points move off the pixel's ray
        r.render(); r.render()
        r.blur_value = 0.0
        r.render()
        outs += all_buffers(r)
        r.destroy(); sc.gvol.destroy()
        return outs

    a, b = run(0), run(1)
    for k, (x, y) in enumerate(zip(a, b)):
        assert_same_bits(y, x, "output %d" % k)


def test_option_toggles_between_passes(gpu_ctx, oracle):
    """fast-math, the atlas, the classes themselves and the filter switched between passes of one sequence: the MISS tiles'
    positions are materialised in the arithmetic that produced their directions"""
    sc = far_scene(gpu_ctx, oracle)

    def run(classes):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.extinction = 4; r.steps = 3
        r.reset()
        seq = [(N.OPTION_FAST_MATH, 1), (N.OPTION_FAST_MATH, 0), (N.OPTION_BOUNDARY_ATLAS, 0), (N.OPTION_FAST_MATH, 1),
               (N.OPTION_BOUNDARY_ATLAS, 1), (N.OPTION_SPLIT_STREAMS, 3), (N.OPTION_FAST_MATH, 0), (N.OPTION_SPLIT_STREAMS, 1)]
        for opt, val in seq:
            r.render(); r.render()
            r.set_option(opt, val)
        if classes:
            r.set_option(N.OPTION_TILE_CLASSES, 0)
            r.render()
            r.set_option(N.OPTION_TILE_CLASSES, 1)
        else:
            r.render()
        r.render()
        outs = all_buffers(r)
        r.destroy()
        return outs

    for k, (x, y) in enumerate(zip(run(0), run(1))):
        assert_same_bits(y, x, "output %d" % k)
    sc.gvol.destroy()


@pytest.mark.parametrize("world,rows", [(3, 5), (8, 8), (2, 16)])
@pytest.mark.parametrize("fast", [0, 1])
def test_sharded_classes_equal_the_unsharded_general_kernel(gpu_ctx, oracle, world, rows, fast):
    sc = far_scene(gpu_ctx, oracle, w=200, h=230)
    whole = sc.renderer('mcm')
    whole.set_option(N.OPTION_TILE_CLASSES, 0)
    whole.set_option(N.OPTION_FAST_MATH, fast)
    whole.extinction = 4; whole.steps = 4
    whole.reset()
    for _ in range(4):
        whole.render()
    want = all_buffers(whole)
    whole.destroy()
    for rank in range(world):
        r = sc.renderer('mcm', shard=(rank, world, rows))
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.extinction = 4; r.steps = 4
        r.reset()
        for _ in range(4):
            r.render()
        g = r.global_rows()
        ok = g >= 0
        for k, (x, y) in enumerate(zip(all_buffers(r), want)):
            assert_same_bits(x[ok], y[g[ok]], "rank %d of %d, output %d" % (rank, world, k))
        assert r.tile_classes()[2] == 0
        r.destroy()
    sc.gvol.destroy()


@pytest.mark.parametrize("fast", [0, 1])
def test_frame_sequences_after_classified_passes(gpu_ctx, oracle, fast):
    """vpt_renderer_play in every mode mixed with classified render() calls"""
    sc = far_scene(gpu_ctx, oracle)

    def run(classes):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.extinction = 4; r.steps = 4
        r.reset()
        r.render(); r.render()
        r.play(3, use_graph=False)
        r.render()
        r.play(3, use_graph=True); r.play(3, use_graph=True)
        r.render()
        r.play(4, fused=True)
        r.render()
        r.play(5, frames=True)
        slots = [r.read_frame_slot(k).copy() for k in range(5)]
        r.render(); r.render()
        outs = all_buffers(r) + slots
        assert r.sample_count() == sc.w * sc.h * 4 * (2 + 3 + 1 + 6 + 1 + 4 + 1 + 5 + 2)
        r.destroy()
        return outs

    for k, (x, y) in enumerate(zip(run(0), run(1))):
        assert_same_bits(y, x, "output %d" % k)
    sc.gvol.destroy()


FORMATS = ["r8-nearest", "rg8-linear", "rg8-nearest", "r32f-linear", "r32f-nearest", "rg32f-linear", "rg32f-nearest"]


def format_scene(gpu_ctx, oracle, fmt, w=208, h=144):
    """far_scene with a volume of another texel format / filter (Volume.js:115-125 setFilter; RAWReader.js:36-38 and the BVP manifests' formats):
    two channels feed a 2-D transfer function, float texels reach outside [0, 1]"""
    kind, filt = fmt.split("-")
    rng = np.random.default_rng(7)
    base = sphere_volume(0, noise=35.0, dims=(20, 26, 22))
    if kind == "r8":
        vol = base
    elif kind == "rg8":
        vol = np.stack([base, rng.integers(0, 256, size=base.shape, dtype=np.uint8)], axis=-1)
    else:
        f = (base.astype(np.float32) / np.float32(255) * np.float32(1.4) - np.float32(0.2)).astype(np.float32)
        vol = f if kind == "r32f" else np.stack([f, rng.uniform(-0.2, 1.2, size=base.shape).astype(np.float32)], axis=-1)
    tf = rng.integers(0, 256, size=(7, 24, 4), dtype=np.uint8) if vol.ndim == 4 else colour_tf(48, 1)
    sc = Scene.__new__(Scene)
    sc.vol, sc.w, sc.h, sc.tf, sc.env = vol, w, h, tf, None
    sc.osc = oracle.OracleScene(vol, filt, tf=tf)
    sc.gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, filt)
    sc.camera = orbit_camera(w / h, 0.7, -0.3, 3.2)
    from vpt_amd.scene import Transform, Node, mvp_inverse_matrix
    sc.transform = Transform(Node())
    sc.m = mvp_inverse_matrix(sc.camera, sc.transform)
    sc.ctx = gpu_ctx
    return sc


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("fast", [0, 1])
def test_classes_for_every_volume_format(gpu_ctx, oracle, fmt, fast):
    """round 4: NEAREST filter, two-channel and float volumes run the tile classes too — the MISS tiles through k_mcm_miss's sampler of the
    volume's boundary atlas in that format (one dword or one float4 per cell and channel), the HIT tiles through the general kernel of the
    volume's variant from a tile list.  Classes on (the default) = classes off, bit for bit, both arithmetic variants; contract arithmetic =
    the oracle; no event of a MISS tile inside the cube"""
    sc = format_scene(gpu_ctx, oracle, fmt)

    def run(classes):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.extinction = 4; r.steps = 5; r.anisotropy = 0.25
        r.reset()
        if classes:
            hit, miss, _ = r.tile_classes()
            assert miss > hit > 0, (hit, miss)
            r.set_profiling(1)
        outs = []
        for k in range(5):
            r.render()
            if k in (0, 4):
                outs += all_buffers(r)
        if classes:
            assert r.profile_side()[1] == 5, "the MISS-tile kernel did not run"         # five passes put a launch on the side stream
            r.set_profiling(False)
        assert r.tile_classes()[2] == 0
        r.destroy()
        return outs

    a, b = run(0), run(1)
    for k, (x, y) in enumerate(zip(a, b)):
        assert_same_bits(y, x, "%s: tile classes on vs off, output %d" % (fmt, k))
    if not fast:
        r = sc.renderer('mcm')
        r.extinction = 4; r.steps = 5; r.anisotropy = 0.25
        o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
        r.reset()
        o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
        for _ in range(3):
            r.render()
            o.render(to_frame(oracle, sc, r._u))
        for bb, st in zip(MCM_BUFFERS, o.state):
            assert_same_bits(r.read(bb), st.reshape(sc.h, sc.w, 4), "%s: state buffer %d against the oracle" % (fmt, bb))
        r.destroy()
    sc.gvol.destroy()


@pytest.mark.parametrize("fmt", ["r8-linear"] + FORMATS)
def test_the_boundary_atlas_gives_the_bricks_samples_in_every_format(gpu_ctx, oracle, fmt):
    """the sample a MISS tile executes and discards is not visible in any buffer — so the atlas sampler itself is probed: texture(uVolume, p) ->
    transfer function at 40 000 positions outside the cube (one, two and three coordinates out of range, far away, on the faces' edges, +-inf)
    through the boundary atlas against the same positions through the bricks, bit for bit"""
    sc = format_scene(gpu_ctx, oracle, fmt) if fmt != "r8-linear" else far_scene(gpu_ctx, oracle)
    r = sc.renderer('mcm')
    rng = np.random.default_rng(3)
    p = rng.uniform(-0.6, 1.6, size=(40000, 3)).astype(np.float32)
    out = ((p < 0) | (p > 1)).any(axis=1)
    p = p[out]
    p[:200] *= np.float32(40.0)                                  # far away
    p[200:400, 0] = np.float32(1.0) + np.float32(1e-7)           # a hair beyond a face
    p[400:500, 1] = np.inf; p[500:600, 2] = -np.inf
    edge = rng.choice([0.0, 1.0, 0.5 / 22, 1 - 0.5 / 22], size=(300, 2)).astype(np.float32)
    p[600:900, 1:] = edge                                        # on texel centres and face edges of the in-face axes
    p[600:900, 0] = np.float32(1.25)
    got, want = r.probe_sample_boundary(p), r.probe_sample(p)
    assert_same_bits(got, want, "%s: atlas vs bricks at %d out-of-cube positions" % (fmt, len(p)))
    r.destroy(); sc.gvol.destroy()


def test_a_float_volume_with_non_finite_texels_keeps_the_general_kernel(gpu_ctx, oracle):
    """lerp(t, t', 0) = fma(0, t' - t, t) is t only while t' - t is finite: a float volume holding inf / NaN / huge texels does not use its
    boundary atlas (vpt_volume_finalize scans it), so nothing is classified away — and the images still equal the oracle's"""
    base = sphere_volume(0, noise=35.0, dims=(16, 20, 18)).astype(np.float32) / np.float32(255)
    base[0, 3, 4] = np.inf; base[15, 7, 2] = np.float32(3e38); base[5, 0, 9] = -np.inf
    sc = format_scene(gpu_ctx, oracle, "r32f-linear")
    sc.gvol.destroy()
    sc.vol = base
    sc.osc = oracle.OracleScene(base, "linear", tf=sc.tf)
    sc.gvol = vpt_amd.Volume.from_array(gpu_ctx, base, "linear")
    r = sc.renderer('mcm')
    r.extinction = 4; r.steps = 4
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset()
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    r.set_profiling(1)
    for _ in range(3):
        r.render()
        o.render(to_frame(oracle, sc, r._u))
    assert r.profile_side()[1] == 0                                      # no MISS-tile kernel ran: its sampler reads the atlas
    with pytest.raises(vpt_amd.VptError):
        r.probe_sample_boundary(np.float32([[1.5, 0.5, 0.5]]))
    for bb, st in zip(MCM_BUFFERS, o.state):
        assert_same_bits(r.read(bb), st.reshape(sc.h, sc.w, 4), "state buffer %d" % bb)
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("fast,form", [(0, 0), (1, 0), (1, 2), (0, 2)])
def test_full_hd_frame_classes_on_and_off(gpu_ctx, oracle, fast, form):
    """the benchmark's image size and camera (1920x1080, default camera: ~77 % MISS tiles) on a 128^3 volume, three streams; both
    forms of the HIT-tile kernel"""
    from vpt_amd.scene import default_camera
    sc = Scene(gpu_ctx, oracle, 128, 1920, 1080, camera=default_camera(1920 / 1080), noise=48.0)

    def run(classes):
        r = mcm_with_form(sc, form)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, 3)
        r.reset()
        if classes:
            hit, miss, _ = r.tile_classes()
            assert hit + miss == 120 * 68 and miss > 0.7 * (hit + miss)
        for _ in range(12):
            r.render()
        outs = all_buffers(r)
        assert r.sample_count() == 1920 * 1080 * 8 * 12
        assert r.tile_classes()[2] == 0
        r.destroy()
        return outs

    for k, (x, y) in enumerate(zip(run(0), run(1))):
        assert_same_bits(y, x, "output %d" % k)
    sc.gvol.destroy()


def _frame_ring(r):
    import ctypes as C
    p, n = C.c_void_p(), C.c_size_t()
    N.check(N.lib().vpt_renderer_frame_ring_device(r._h, C.byref(p), C.byref(n)))
    return p.value, n.value


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("shard,form", [(None, 0), ((1, 3, 8), 0), (None, 1)])
def test_bucket_kernel_equals_frame_by_frame(gpu_ctx, oracle, fast, shard, form):
    """VPT_OPTION_BUCKET_KERNEL: vpt_renderer_play_into runs its frames by one launch per tile class (k_mcm_bucket_hit | k_mcm_bucket_miss,
    seeds by value, state in registers).  Every frame of every bucket, all four state buffers and the sample count identical to the
    frame-by-frame form; buckets of 1, 7 and 16 frames and one of 21 (two launches), render() calls in between, a camera that moves
    without a reset (the classes are void: frame by frame from there on) and a second reset.  The renderer's own frame ring stands in
    for the caller's bucket memory."""
    env = env_map(16, 8)

    def run(bucket):
        sc = far_scene(gpu_ctx, oracle, w=200, h=184, env=env)
        r = mcm_with_form(sc, form, shard=shard) if shard else mcm_with_form(sc, form)
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_BUCKET_KERNEL, bucket)
        r.extinction = 4; r.steps = 4; r.anisotropy = 0.2
        r.reset()
        r.render()
        r.play(16, frames=True)                             # allocates the ring (and lets read_frame_slot reach all 16 slots)
        ring, slot = _frame_ring(r)
        outs = []

        def bucket_of(n):
            done = 0
            while done < n:                                 # the ring holds 16 slots: 21 frames = 16 + 5
                m = min(16, n - done)
                r.play_into(m, ring, slot)
                for k in (0, m // 2, m - 1):
                    outs.append(r.read_frame_slot(k).copy())
                done += m

        bucket_of(1); bucket_of(7)
        r.set_render_target(0, 0)
        r.render(); r.render()
        outs.extend(all_buffers(r))
        bucket_of(16); bucket_of(21)
        r.set_render_target(0, 0)
        outs.extend(all_buffers(r))
        sc.camera.transform.localTranslation = [0.3, 0.2, 1.6]       # no reset: the classes are void, the bucket goes frame by frame
        sc.camera.transform.localRotation = [0, 0, 0, 1]
        bucket_of(5)
        r.set_render_target(0, 0)
        r.reset()
        bucket_of(6)
        r.set_render_target(0, 0)
        r.render()
        outs.extend(all_buffers(r))
        count = r.sample_count()
        # buckets of 1, 7, 16, 16 + 5 and (after the second reset) 6 frames went through the bucket kernels; the 5 behind the moved camera did not
        assert r.bucket_launches() == (6 if bucket else 0) + 1      # (+ the play(16, frames=True) above: VPT_PLAY_FRAMES runs them wherever the classes are in force)
        if not shard:
            assert count == sc.w * sc.h * 4 * (1 + 16 + 1 + 7 + 2 + 16 + 21 + 5 + 6 + 1)
        r.destroy(); sc.gvol.destroy()
        return outs, count

    (a, na), (b, nb) = run(0), run(1)
    assert len(a) == len(b) and na == nb
    for k, (x, y) in enumerate(zip(a, b)):
        assert_same_bits(y, x, "bucket kernel vs frame by frame, output %d" % k)


def test_a_default_constructed_renderer_runs_the_tile_classes(gpu_ctx, oracle):
    """round 4: the measured best form is the library's default — RendererFactory('mcm') + reset() + render() with NO option call launches the
    HIT | MISS kernels on two streams (VPT_OPTION_SPLIT_STREAMS defaults to 2 for MCM, the side stream is created by the first pass)"""
    sc = far_scene(gpu_ctx, oracle)
    r = sc.renderer('mcm')
    r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)          # (counts only; does not change what is launched)
    r.extinction = 5
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset()
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    hit, miss, _ = r.tile_classes()
    assert miss > 0 and hit > 0
    r.set_profiling(1)
    for _ in range(3):
        r.render()
        o.render(to_frame(oracle, sc, r._u))
    assert r.profile_side()[1] == 3                        # three passes put a launch on the side stream: the MISS-tile kernel
    r.set_profiling(False)
    for b, s in zip(MCM_BUFFERS, o.state):
        assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "state buffer %d" % b)
    assert r.tile_classes()[2] == 0
    r.destroy(); sc.gvol.destroy()


def test_a_distant_camera_runs_every_tile_as_a_hit_tile(gpu_ctx, oracle):
    """near-plane points thousands of units from the cube: the classification's absolute margin no longer covers the kernels' fp32 rounding
    there, every tile is HIT (vpt_core.hip VPT_CLASS_FAR) — and the images agree with the classes switched off, VERIFY on"""
    from vpt_amd.scene import default_camera
    w, h = 208, 144
    cam = default_camera(w / h)
    cam.transform.localTranslation = [0.3, 0.2, 5000.0]
    pc = cam.components[0]
    pc.fovy = 0.0006; pc.near = 2500.0; pc.far = 10000.0
    sc = Scene(gpu_ctx, oracle, 24, w, h, tf=colour_tf(48, 1), camera=cam)
    outs = []
    for classes in (0, 1):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.extinction = 3
        r.reset()
        if classes:
            assert r.tile_classes()[1] == 0
        for _ in range(3):
            r.render()
        assert r.tile_classes()[2] == 0
        outs.append(all_buffers(r))
        r.destroy()
    for k, (x, y) in enumerate(zip(*outs)):
        assert_same_bits(y, x, "distant camera, output %d" % k)
    sc.gvol.destroy()


def test_resets_with_moving_and_resting_cameras_reuse_or_rebuild_the_lists(gpu_ctx, oracle):
    """round 4: the class lists are cached per (matrix, geometry) and otherwise uploaded from two pinned staging buffers in turn without a host
    wait.  Seven resets — the camera moves, rests, moves back, the image is resized — each followed by passes: classes on = classes off"""
    cams = [(0.7, -0.3, 3.2), (0.7, -0.3, 3.2), (1.9, 0.4, 2.6), (-0.8, 0.2, 3.0), (0.7, -0.3, 3.2), (0.7, -0.3, 3.2), (2.5, -0.5, 2.2)]

    def run(classes):
        sc = far_scene(gpu_ctx, oracle)
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.extinction = 4; r.steps = 4
        outs, counts = [], []
        for k, cam in enumerate(cams):
            if k == 5:
                r.setResolution((176, 128))                 # new geometry: the cached lists are void
            c = orbit_camera((176 / 128) if k >= 5 else (sc.w / sc.h), *cam)
            r._camera = c
            r.reset()
            counts.append(r.tile_classes()[:2])
            r.render(); r.render()
            outs.append(r.getTexture().copy())
            outs.append(r.read(N.BUFFER_MCM_RADIANCE).copy())
        assert r.tile_classes()[2] == 0
        r.destroy(); sc.gvol.destroy()
        return outs, counts

    (a, ca), (b, cb) = run(0), run(1)
    assert cb[0] == cb[1] == cb[4] and cb[2] != cb[0]          # the same camera gives the same lists; another camera others
    for k, (x, y) in enumerate(zip(a, b)):
        assert_same_bits(y, x, "output %d" % k)
