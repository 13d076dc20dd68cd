"""GPU: the ISO and Depth renderers against the CPU oracle, bit for bit: every pass hook by
hook and fused, frame / accumulation / render buffers, sample counts, filters, sharding."""
import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import Transform, Node, default_camera, mvp_inverse_matrix, iso_light_direction
from vpt_amd.synthetic import sphere_volume, colour_tf, GoldenRatioRng

from conftest import orbit_camera

pytestmark = pytest.mark.gpu


def same_bits(got, want, what):
    g = np.ascontiguousarray(got).view(np.uint8).reshape(-1); w = np.ascontiguousarray(want).view(np.uint8).reshape(-1)
    assert g.shape == w.shape, what
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, "%s: %d of %d bytes differ, first at byte %d" % (what, bad.size, g.size, bad[0])


class Scene:
    def __init__(self, ctx, oracle, n, w, h, filt="linear", tf=None, camera=None, dims=None):
        self.vol = sphere_volume(n, noise=40.0, dims=dims)
        self.w, self.h, self.tf, self.ctx = w, h, tf, ctx
        self.osc = oracle.OracleScene(self.vol, filt, tf=tf)
        self.gvol = vpt_amd.Volume.from_array(ctx, self.vol, filt)
        self.camera = camera if camera is not None else default_camera(w / h)
        self.transform = Transform(Node())
        self.m = mvp_inverse_matrix(self.camera, self.transform)

    def renderer(self, kind, **opts):
        o = {'resolution': (self.w, self.h), 'transform': self.transform, 'rng': GoldenRatioRng()}
        o.update(opts)
        r = vpt_amd.RendererFactory(kind)(self.ctx, self.gvol, self.camera, None, o)
        if self.tf is not None:
            r.setTransferFunction(self.tf)
        return r


@pytest.mark.parametrize("filt", ["linear", "nearest"])
@pytest.mark.parametrize("fused", [False, True])
def test_iso_parity(gpu_ctx, oracle, filt, fused):
    sc = Scene(gpu_ctx, oracle, 40, 136, 104, filt, tf=colour_tf(64, 1), camera=orbit_camera(136 / 104), dims=(37, 40, 33))
    r = sc.renderer('iso', fused=fused)
    r.steps = 37                                    # 1/37 is not exact; also not a multiple of the kernel's unroll of 4
    r.isovalue = 0.3
    r.light = [1.5, -2.0, -3.0]
    o = oracle.OracleRenderer('iso', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    rng = GoldenRatioRng()
    light = iso_light_direction(sc.camera, sc.transform, r.light)
    for k in range(3):
        r.render()
        fr = oracle.make_frame(sc.w, sc.h, sc.m, offset=np.float32(rng()), mcm_steps=37, isovalue=0.3, light_dir=light, gradient_step=0.005)
        fr.step = float(np.float32(1.0) / np.float32(37))
        o.render(fr)
        if not fused:
            same_bits(r.read(N.BUFFER_FRAME).view(np.uint16), o.frame.reshape(sc.h, sc.w, 4), "iso frame %d" % k)
        same_bits(r.read(N.BUFFER_ACCUM).view(np.uint16), o.acc.reshape(sc.h, sc.w, 4), "iso closest %d" % k)
        same_bits(r.getTexture().view(np.uint16), o.out.reshape(sc.h, sc.w, 4), "iso render %d" % k)
    assert r.sample_count() == o.samples and o.samples > 0
    acc = r.read(N.BUFFER_ACCUM).astype(np.float32)
    assert (acc[..., 3] > 0).any() and (acc[..., 3] == -1).any()        # hits and misses both present
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("random", [False, True])
@pytest.mark.parametrize("fused", [False, True])
def test_depth_parity(gpu_ctx, oracle, fused, random):
    sc = Scene(gpu_ctx, oracle, 36, 120, 88, tf=colour_tf(64, 1), camera=orbit_camera(120 / 88))
    r = sc.renderer('depth', fused=fused)
    r.slices = 45; r.extinction = 70; r.threshold = 0.2; r.random = random
    o = oracle.OracleRenderer('depth', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    rng = GoldenRatioRng()
    for k in range(3):
        r.render()
        off = np.float32(rng()) if random else 0.0
        fr = oracle.make_frame(sc.w, sc.h, sc.m, offset=off, steps=45, extinction=70, threshold=0.2, mix=np.float32(1.0 / (k + 1)))
        o.render(fr)
        if not fused:
            same_bits(r.read(N.BUFFER_FRAME), o.frame.reshape(sc.h, sc.w), "depth frame %d" % k)
        same_bits(r.read(N.BUFFER_ACCUM), o.acc.reshape(sc.h, sc.w), "depth accumulation %d" % k)
        same_bits(r.getTexture().view(np.uint16), o.out.reshape(sc.h, sc.w, 4), "depth render %d" % k)
    assert r.sample_count() == o.samples and o.samples > 0
    d = r.read(N.BUFFER_ACCUM)
    assert (d > 0).any() and (d < 0).any()
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("kind", ["iso", "depth"])
def test_sharded_equals_unsharded(gpu_ctx, oracle, kind):
    sc = Scene(gpu_ctx, oracle, 32, 100, 75, tf=colour_tf(64, 1))

    def run(**opts):
        r = sc.renderer(kind, **opts)
        r.reset()
        for _ in range(2):
            r.render()
        out = (r.getTexture(), r.global_rows(), r.sample_count())
        r.destroy()
        return out

    whole, _, ns = run()
    got = np.zeros_like(whole); total = 0
    for rank in range(3):
        img, rows, n = run(shard=(rank, 3, 8))
        got[rows[rows >= 0]] = img[rows >= 0]; total += n
    same_bits(got.view(np.uint16), whole.view(np.uint16), "%s sharded" % kind)
    assert total == ns
    sc.gvol.destroy()


def test_properties_and_reset_rules(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 16, 32, 32)
    iso = sc.renderer('iso')
    assert [(p['name'], p['value']) for p in iso.properties[:3]] == [('steps', 50), ('isovalue', 0.5), ('light', [2, -3, -5])]   # ISORenderer.js:17-40
    dep = sc.renderer('depth')
    assert [(p['name'], p['value']) for p in dep.properties[:4]] == [('extinction', 100), ('slices', 64), ('threshold', 0.1), ('random', False)]   # DepthRenderer.js:17-47
    iso.reset(); iso.render()
    from vpt_amd.property_bag import CustomEvent
    before = iso.read(N.BUFFER_ACCUM).copy()
    iso.dispatchEvent(CustomEvent('change', {'detail': {'name': 'steps', 'value': 10}}))        # not in the reset list (ISORenderer.js:55-60)
    same_bits(iso.read(N.BUFFER_ACCUM), before, "steps change must not reset")
    iso.dispatchEvent(CustomEvent('change', {'detail': {'name': 'isovalue', 'value': 0.6}}))
    assert (iso.read(N.BUFFER_ACCUM).view(np.uint16) == 0xbc00).all()                             # reset: vec4(-1)
    dep.reset(); dep.render(); dep.render()
    assert dep._frameNumber == 2
    dep.dispatchEvent(CustomEvent('change', {'detail': {'name': 'threshold', 'value': 0.3}}))
    assert dep._frameNumber == 0 and (dep.read(N.BUFFER_ACCUM) == 0).all()
    u = iso._prepare_generate(); u.steps = 0
    with pytest.raises(vpt_amd.VptError, match="steps"):
        N.check(N.lib().vpt_renderer_generate(iso._h, __import__('ctypes').byref(u)))
    iso.destroy(); dep.destroy(); sc.gvol.destroy()


def test_full_size_oracle_bands(gpu_ctx, oracle):
    """1920x1080 on a 256^3 volume: fused == hooks and a 12-row oracle band for both renderers"""
    sc = Scene(gpu_ctx, oracle, 256, 1920, 1080)
    y0, y1 = 534, 546
    for kind in ("iso", "depth"):
        outs = []
        for fused in (True, False):
            r = sc.renderer(kind, fused=fused)
            r.reset()
            for _ in range(2):
                r.render()
            outs.append((r.getTexture(), r.read(N.BUFFER_ACCUM), r.sample_count()))
            light = iso_light_direction(sc.camera, sc.transform, r.light) if kind == "iso" else None
            r.destroy()
        same_bits(outs[0][0].view(np.uint16), outs[1][0].view(np.uint16), "%s fused vs hooks" % kind)
        assert outs[0][2] == outs[1][2]
        o = oracle.OracleRenderer(kind, sc.osc, sc.w, sc.h)
        rng = GoldenRatioRng()
        o.reset(oracle.make_frame(sc.w, sc.h, sc.m, y0=y0, y1=y1))
        for k in range(2):
            if kind == "iso":
                fr = oracle.make_frame(sc.w, sc.h, sc.m, offset=np.float32(rng()), mcm_steps=50, isovalue=0.5, light_dir=light, y0=y0, y1=y1, nthreads=8)
                fr.step = float(np.float32(1.0) / np.float32(50))
            else:
                fr = oracle.make_frame(sc.w, sc.h, sc.m, offset=0.0, steps=64, extinction=100, threshold=0.1, mix=np.float32(1.0 / (k + 1)), y0=y0, y1=y1, nthreads=8)
            o.render(fr)
        same_bits(outs[0][0][y0:y1].view(np.uint16), o.out.reshape(sc.h, sc.w, 4)[y0:y1], "%s oracle band" % kind)
    sc.gvol.destroy()
