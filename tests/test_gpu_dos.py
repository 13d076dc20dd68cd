"""GPU: the DOS renderer (src/js/renderers/DOSRenderer.js, src/glsl/renderers/DOSRenderer.glsl)
against the CPU oracle, bit for bit, slice by slice: colour and occlusion buffers, render buffer, sample counts, the
progressive sweep over several render() calls, REPEAT wrap of the occlusion taps, filters, RG8 volumes.
Parity unpinned by the reference itself: it holds no output fixture for this renderer."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import Transform, Node, default_camera, mvp_inverse_matrix
from vpt_amd.synthetic import sphere_volume, colour_tf, GoldenRatioRng

from conftest import orbit_camera

pytestmark = pytest.mark.gpu


def same_bits(got, want, what):
    g = np.ascontiguousarray(got).view(np.uint8).reshape(-1); w = np.ascontiguousarray(want).view(np.uint8).reshape(-1)
    assert g.shape == w.shape, what
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, "%s: %d of %d bytes differ, first at byte %d" % (what, bad.size, g.size, bad[0])


class Scene:
    def __init__(self, ctx, oracle, vol, w, h, filt="linear", tf=None, camera=None):
        self.vol = vol
        self.w, self.h, self.tf, self.ctx = w, h, tf, ctx
        self.osc = oracle.OracleScene(vol, filt, tf=tf)
        self.gvol = vpt_amd.Volume.from_array(ctx, vol, filt)
        self.camera = camera if camera is not None else default_camera(w / h)
        self.transform = Transform(Node())
        self.m = mvp_inverse_matrix(self.camera, self.transform)

    def renderer(self, **opts):
        o = {'resolution': (self.w, self.h), 'transform': self.transform, 'rng': GoldenRatioRng()}
        o.update(opts)
        r = vpt_amd.DOSRenderer(self.ctx, self.gvol, self.camera, None, o)
        if self.tf is not None:
            r.setTransferFunction(self.tf)
        return r


def sweep(sc, oracle, r, calls, what, nthreads=4):
    """reset, then `calls` render() calls, every buffer compared after each"""
    o = oracle.OracleRenderer('dos', sc.osc, sc.w, sc.h)
    r.reset()
    fr = oracle.make_frame(sc.w, sc.h, sc.m, nthreads=nthreads)
    o.reset(fr)
    same_bits(r.read(N.BUFFER_ACCUM), o.color[o.cur], what + " reset colour")
    same_bits(r.read(N.BUFFER_DOS_OCCLUSION), o.occlusion[o.cur], what + " reset occlusion")
    total = 0
    for k in range(calls):
        r.render()
        u = r._u
        fr = oracle.make_frame(sc.w, sc.h, np.array(list(u.mvp_inverse), np.float32), extinction=u.extinction, nthreads=nthreads)
        fr.step = u.step_size
        o.integrate_slices(fr, r._slices, r._occlusionSamples)
        o.render_frame(fr)
        total += len(r._slices)
        same_bits(r.read(N.BUFFER_ACCUM), o.color[o.cur], "%s colour after call %d" % (what, k))
        same_bits(r.read(N.BUFFER_DOS_OCCLUSION), o.occlusion[o.cur], "%s occlusion after call %d" % (what, k))
        same_bits(r.getTexture().view(np.uint16), o.out, "%s render after call %d" % (what, k))
    assert r.sample_count() == o.samples, what
    return o, total


@pytest.mark.parametrize("filt", ["linear", "nearest"])
def test_dos_parity_progressive_sweep(gpu_ctx, oracle, filt):
    sc = Scene(gpu_ctx, oracle, sphere_volume(40, noise=40.0, dims=(37, 40, 33)), 104, 80, filt, tf=colour_tf(64, 1), camera=orbit_camera(104 / 80))
    r = sc.renderer()
    r.slices = 45; r.steps = 20; r.extinction = 70; r.aperture = 40
    o, total = sweep(sc, oracle, r, 4, "dos " + filt)               # 20 + 20 + 5 or 6 (the sweep ends past the far corner) + 0
    assert total in (45, 46) and len(r._slices) == 0          # depth_min + 45 * sliceDistance lands on the far corner, give or take an ulp
    col = o.color[o.cur].reshape(sc.h, sc.w, 4); occ = o.occlusion[o.cur]
    assert (col[..., 3] > 0.5).any() and (col[..., 3] == 0).any() and (occ < 0.9).any() and (occ == 1.0).any() and o.samples > 0
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("case", [
    dict(samples=1, aperture=0),                       # a single tap with no extent: the texel itself
    dict(samples=37, aperture=60, extinction=300),
    dict(samples=8, aperture=89, slices=7, steps=3),   # tan(89 deg): taps wrap around the image several times (REPEAT)
    dict(samples=5, aperture=30, extinction=0),
    dict(samples=200, aperture=10, slices=12, steps=12),
])
def test_dos_parameter_switches(gpu_ctx, oracle, case):
    sc = Scene(gpu_ctx, oracle, sphere_volume(24, noise=40.0), 72, 56, tf=colour_tf(32, 1), camera=orbit_camera(72 / 56))
    r = sc.renderer()
    r.slices = 20; r.steps = 8
    for k, v in case.items():
        setattr(r, k, v)
    r.generateOcclusionSamples()                       # property writes do not dispatch 'change' (PropertyBag.js)
    assert len(r._occlusionSamples) == 2 * r.samples
    sweep(sc, oracle, r, 2, str(case))
    r.destroy(); sc.gvol.destroy()


def test_dos_rg8_volume_and_2d_transfer_function(gpu_ctx, oracle):
    rng = np.random.default_rng(5)
    vol = np.ascontiguousarray(np.stack([sphere_volume(28, noise=30.0), rng.integers(0, 256, size=(28, 28, 28), dtype=np.uint8)], axis=-1))
    tf = rng.integers(0, 256, size=(5, 16, 4), dtype=np.uint8)
    sc = Scene(gpu_ctx, oracle, vol, 90, 61, tf=tf, camera=orbit_camera(90 / 61))
    r = sc.renderer()
    r.slices = 30; r.steps = 16; r.extinction = 40
    sweep(sc, oracle, r, 2, "dos rg8")
    r.destroy(); sc.gvol.destroy()


def test_dos_camera_inside_the_volume_and_odd_sizes(gpu_ctx, oracle):
    cam = default_camera(33 / 47)
    cam.transform.localTranslation = [0.1, -0.05, 0.3]             # inside the cube: the sweep starts at depth 0
    sc = Scene(gpu_ctx, oracle, sphere_volume(20, noise=60.0), 33, 47, tf=colour_tf(16, 1), camera=cam)
    r = sc.renderer()
    r.slices = 25; r.steps = 30; r.extinction = 20
    r.reset()
    assert r._minDepth == 0 and r._maxDepth > 0
    sweep(sc, oracle, r, 1, "dos inside")
    r.destroy(); sc.gvol.destroy()


def test_dos_properties_reset_rules_and_errors(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, sphere_volume(16), 32, 32)
    r = sc.renderer()
    want = [('steps', 50), ('slices', 200), ('extinction', 100), ('aperture', 30), ('samples', 8)]
    assert [(p['name'], p['value']) for p in r.properties[:5]] == want            # DOSRenderer.js:18-57
    s = r._occlusionSamples.reshape(-1, 2).astype(np.float64)
    assert s.shape == (8, 2) and abs(s.mean(axis=0)).max() < 1e-6                 # centred (:121-124)
    r.reset(); r.render()
    from vpt_amd.property_bag import CustomEvent
    before = r.read(N.BUFFER_ACCUM).copy()
    assert (before[..., 3] > 0).any()
    r.dispatchEvent(CustomEvent('change', {'detail': {'name': 'steps', 'value': 10}}))          # not in the reset list (:76-82)
    same_bits(r.read(N.BUFFER_ACCUM), before, "steps change must not reset")
    old = r._occlusionSamples.copy()
    r.samples = 5
    r.dispatchEvent(CustomEvent('change', {'detail': {'name': 'samples', 'value': 5}}))         # regenerates and resets (:72-84)
    assert len(r._occlusionSamples) == 10 and not np.array_equal(r._occlusionSamples, old[:10])
    assert (r.read(N.BUFFER_ACCUM) == 0).all() and (r.read(N.BUFFER_DOS_OCCLUSION) == 1).all()  # DOSRenderer.glsl:141-143
    u = r._new_uniforms()
    with pytest.raises(vpt_amd.VptError, match="integrate_slices"):
        N.check(N.lib().vpt_renderer_integrate(r._h, C.byref(u)))
    with pytest.raises(vpt_amd.VptError, match="single-launch"):
        N.check(N.lib().vpt_renderer_render(r._h, C.byref(u)))
    with pytest.raises(vpt_amd.VptError, match="does not shard"):
        N.check(N.lib().vpt_renderer_set_shard(r._h, 0, 2, 8))
    with pytest.raises(vpt_amd.VptError, match="frame sequences"):
        r.play(2)
    with pytest.raises(vpt_amd.VptError, match="never written"):
        r.read(N.BUFFER_FRAME)
    bad = np.zeros(2, np.float32)
    with pytest.raises(vpt_amd.VptError, match="out of range"):
        N.check(N.lib().vpt_renderer_set_occlusion_samples(r._h, bad.ctypes.data_as(C.c_void_p), 0))
    mip = vpt_amd.MIPRenderer(sc.ctx, sc.gvol, sc.camera, None, {'resolution': (32, 32), 'transform': sc.transform})
    with pytest.raises(vpt_amd.VptError, match="not a DOS renderer"):
        N.check(N.lib().vpt_renderer_integrate_slices(mip._h, C.byref(u), bad.ctypes.data_as(C.c_void_p), 0))
    N.check(N.lib().vpt_renderer_set_shard(r._h, 0, 1, 8))                        # a world of one is fine
    mip.destroy(); r.destroy(); sc.gvol.destroy()


def test_dos_full_size_oracle(gpu_ctx, oracle):
    """1920x1080 on a 128^3 volume, 12 slices of the default sweep: the whole frame against the oracle"""
    sc = Scene(gpu_ctx, oracle, sphere_volume(128, noise=40.0), 1920, 1080, tf=colour_tf(64, 1))
    r = sc.renderer()
    r.slices = 24; r.steps = 12
    sweep(sc, oracle, r, 1, "dos 1080p", nthreads=8)
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("seed", range(*[int(v) for v in os.environ.get("VPT_FUZZ_SEEDS", "0:24").split(":")]))
def test_dos_random_scene(gpu_ctx, oracle, seed):
    """the random scenes of test_gpu_fuzz.py (image / volume shapes down to 1, RG8, NEAREST, cameras inside, outside and
    looking away, scaled / rotated models, 2-D transfer functions) swept by the DOS renderer"""
    from test_gpu_fuzz import random_case, random_camera
    rng, vol, (w, h), tf, env, filt, model = random_case(7000 + seed)
    camera = random_camera(rng, w / h)
    sc = Scene.__new__(Scene)
    sc.vol, sc.w, sc.h, sc.tf, sc.ctx = vol, w, h, tf, gpu_ctx
    sc.osc = oracle.OracleScene(vol, filt, tf=tf)
    sc.gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, filt)
    sc.camera, sc.transform = camera, model
    sc.m = mvp_inverse_matrix(camera, model)
    r = sc.renderer(rng=GoldenRatioRng(int(rng.integers(1, 50))))
    r.slices = int(rng.choice([1, 6, 17, 40])); r.steps = int(rng.choice([1, 5, 50])); r.extinction = float(rng.choice([0.0, 10.0, 100.0, 1000.0]))
    r.aperture = float(rng.choice([0, 10, 30, 75, 89])); r.samples = int(rng.choice([1, 2, 8, 33]))
    r.generateOcclusionSamples()
    sweep(sc, oracle, r, 3, "dos seed %d (%dx%d image, volume %s, %s)" % (seed, w, h, vol.shape, filt), nthreads=2)
    r.destroy(); sc.gvol.destroy()


def test_dos_matrix_changes_mid_sweep_and_volume_partly_off_screen(gpu_ctx, oracle):
    """the native side only launches the tiles of the volume's screen bounding box; moving the camera between render() calls
    without a reset (the C-ABI allows it) must still give the oracle's buffers everywhere, and so must a volume that
    leaves the image or is off screen altogether"""
    cam = orbit_camera(160 / 120, dist=3.5)                          # small on screen: most tiles are never launched
    sc = Scene(gpu_ctx, oracle, sphere_volume(24, noise=40.0), 160, 120, tf=colour_tf(32, 1), camera=cam)
    r = sc.renderer()
    r.slices = 30; r.steps = 6; r.extinction = 90
    o = oracle.OracleRenderer('dos', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    moves = [None, [1.2, 0.3, 2.6], [-2.4, -0.2, 2.2], [0.0, 0.0, 60.0], [9.0, 0.0, 2.0], None]     # sideways, far away, off screen
    for k, mv in enumerate(moves):
        if mv is not None:
            cam.transform.localTranslation = mv
        r.render()
        u = r._u
        fr = oracle.make_frame(sc.w, sc.h, np.array(list(u.mvp_inverse), np.float32), extinction=u.extinction, nthreads=4)
        fr.step = u.step_size
        o.integrate_slices(fr, r._slices, r._occlusionSamples); o.render_frame(fr)
        same_bits(r.read(N.BUFFER_ACCUM), o.color[o.cur], "colour after move %d" % k)
        same_bits(r.read(N.BUFFER_DOS_OCCLUSION), o.occlusion[o.cur], "occlusion after move %d" % k)
        same_bits(r.getTexture().view(np.uint16), o.out, "render after move %d" % k)
    assert r.sample_count() == o.samples and o.samples > 0
    r.destroy(); sc.gvol.destroy()
