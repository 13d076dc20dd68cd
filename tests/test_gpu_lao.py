"""GPU: the LAO renderer (src/js/renderers/LAORenderer.js, src/glsl/renderers/LAORenderer.glsl)
against the CPU oracle, bit for bit: hook by hook and fused, every parameter switch, filters, sharding.
Parity unpinned by the reference itself: it holds no output fixture for this renderer."""
import ctypes as C

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
    def __init__(self, ctx, oracle, n, w, h, filt="linear", tf=None, camera=None, dims=None):
        self.vol = sphere_volume(n, noise=40.0, dims=dims)
        self.w, self.h, self.tf, self.ctx = w, h, tf, ctx
        self.osc = oracle.OracleScene(self.vol, filt, tf=tf)
        self.gvol = vpt_amd.Volume.from_array(ctx, self.vol, filt)
        self.camera = camera if camera is not None else default_camera(w / h)
        self.transform = Transform(Node())
        self.m = mvp_inverse_matrix(self.camera, self.transform)

    def renderer(self, **opts):
        o = {'resolution': (self.w, self.h), 'transform': self.transform, 'rng': GoldenRatioRng()}
        o.update(opts)
        r = vpt_amd.LAORenderer(self.ctx, self.gvol, self.camera, None, o)
        if self.tf is not None:
            r.setTransferFunction(self.tf)
        return r


def oracle_params(oracle, r):
    return oracle.lao_params(local_ambient_occlusion=int(bool(r.localAmbientOcclusion)), lao_weight=r.LAOWeight,
                             num_lao_samples=r.numLAOSamples, lao_step_size=r.LAOStepSize, soft_shadows=int(bool(r.softShadows)),
                             shadows_weight=r.shadowsWeight, num_shadow_samples=r.numShadowSamples, light_radius=r.lightRadious,
                             light_coefficient=r.lightCoeficient, light_position=r.lightPosition)


def check(sc, oracle, r, fused, frames=2, what="lao"):
    o = oracle.OracleRenderer('lao', sc.osc, sc.w, sc.h)
    o.lao = oracle_params(oracle, r)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    same_bits(r.read(N.BUFFER_ACCUM), o.acc.reshape(sc.h, sc.w, 4), what + " reset")
    for k in range(frames):
        r.render()
        fr = oracle.make_frame(sc.w, sc.h, sc.m, steps=int(r.slices), extinction=r.extinction)
        o.render(fr)
        if not fused:
            same_bits(r.read(N.BUFFER_FRAME), o.frame.reshape(sc.h, sc.w, 4), "%s frame %d" % (what, k))
        same_bits(r.read(N.BUFFER_ACCUM), o.acc.reshape(sc.h, sc.w, 4), "%s accumulation %d" % (what, k))
        same_bits(r.getTexture().view(np.uint16), o.out.reshape(sc.h, sc.w, 4), "%s render %d" % (what, k))
    assert r.sample_count() == o.samples and o.samples > 0, what
    return r.read(N.BUFFER_ACCUM)


@pytest.mark.parametrize("filt", ["linear", "nearest"])
@pytest.mark.parametrize("fused", [False, True])
def test_lao_parity_defaults(gpu_ctx, oracle, filt, fused):
    sc = Scene(gpu_ctx, oracle, 40, 104, 80, filt, tf=colour_tf(64, 1), camera=orbit_camera(104 / 80), dims=(37, 40, 33))
    r = sc.renderer(fused=fused)
    r.slices = 45; r.extinction = 70
    acc = check(sc, oracle, r, fused)
    assert (acc[..., :3] > 0).any() and (acc[..., :3] == 0).all(axis=-1).any() and (acc[..., 3] == 255).all()
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("case", [
    dict(localAmbientOcclusion=False),
    dict(softShadows=False),
    dict(localAmbientOcclusion=False, softShadows=False),
    dict(numLAOSamples=3, LAOStepSize=0.11, LAOWeight=0.3, lightCoeficient=2.5),
    dict(numShadowSamples=4, shadowsWeight=0.9, lightRadious=0.4, lightPosition=[-3.0, 1.5, 7.0]),
    dict(LAOStepSize=0.002, lightRadious=0.0),
    dict(LAOStepSize=2.0, numLAOSamples=2, lightPosition=[0.5, 0.5, 0.5]),
])
def test_lao_parameter_switches(gpu_ctx, oracle, case):
    sc = Scene(gpu_ctx, oracle, 24, 72, 56, tf=colour_tf(32, 1), camera=orbit_camera(72 / 56))
    r = sc.renderer(fused=False)
    r.slices = 20; r.extinction = 130
    for k, v in case.items():
        setattr(r, k, v)
    check(sc, oracle, r, False, frames=1, what=str(case))
    r.destroy(); sc.gvol.destroy()


def test_lao_sharded_equals_unsharded(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 32, 100, 75, tf=colour_tf(64, 1))

    def run(**opts):
        r = sc.renderer(**opts)
        r.slices = 24
        r.reset()
        r.render()
        out = (r.getTexture(), r.global_rows(), r.sample_count())
        r.destroy()
        return out

    whole, _, ns = run()
    got = np.zeros_like(whole); total = 0
    for rank in range(3):
        img, rows, n = run(shard=(rank, 3, 8))
        got[rows[rows >= 0]] = img[rows >= 0]; total += n
    same_bits(got.view(np.uint16), whole.view(np.uint16), "lao sharded")
    assert total == ns
    sc.gvol.destroy()


def test_lao_properties_reset_rules_and_errors(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 16, 32, 32)
    r = sc.renderer()
    want = [('extinction', 100), ('localAmbientOcclusion', True), ('LAOWeight', 0.69), ('numLAOSamples', 1), ('LAOStepSize', 0.05),
            ('softShadows', True), ('shadowsWeight', 0.54), ('numShadowSamples', 10), ('lightRadious', 0.19), ('lightPosition', [2, 12, 3]),
            ('lightCoeficient', 1.0), ('slices', 64)]
    assert [(p['name'], p['value']) for p in r.properties[:12]] == want          # LAORenderer.js:17-101
    r.slices = 8
    r.reset(); r.render()
    from vpt_amd.property_bag import CustomEvent
    before = r.read(N.BUFFER_ACCUM).copy()
    assert (before[..., :3] > 0).any()
    r.dispatchEvent(CustomEvent('change', {'detail': {'name': 'LAOWeight', 'value': 0.1}}))     # not in the reset list (:116-119)
    same_bits(r.read(N.BUFFER_ACCUM), before, "LAOWeight change must not reset")
    r.dispatchEvent(CustomEvent('change', {'detail': {'name': 'slices', 'value': 9}}))
    acc = r.read(N.BUFFER_ACCUM)
    assert (acc[..., :3] == 0).all() and (acc[..., 3] == 255).all()             # reset: (0, 0, 0, 1), LAORenderer.glsl:285-287
    p = r.lao_params(); p.lao_step_size = 0.0
    with pytest.raises(vpt_amd.VptError, match="step size"):
        N.check(N.lib().vpt_renderer_set_lao_params(r._h, C.byref(p)))
    p = r.lao_params(); p.num_shadow_samples = 0
    with pytest.raises(vpt_amd.VptError, match="sample counts"):
        N.check(N.lib().vpt_renderer_set_lao_params(r._h, C.byref(p)))
    mip = vpt_amd.MIPRenderer(sc.ctx, sc.gvol, sc.camera, None, {'resolution': (32, 32), 'transform': sc.transform})
    with pytest.raises(vpt_amd.VptError, match="not an LAO renderer"):
        N.check(N.lib().vpt_renderer_set_lao_params(mip._h, C.byref(r.lao_params())))
    u = r._prepare_generate(); u.step_size = 0.0
    with pytest.raises(vpt_amd.VptError, match="step"):
        N.check(N.lib().vpt_renderer_generate(r._h, C.byref(u)))
    with pytest.raises(vpt_amd.VptError, match="do not accumulate"):
        r.play(2, fused=True)
    mip.destroy(); r.destroy(); sc.gvol.destroy()


def test_lao_full_size_fused_equals_hooks_and_oracle_band(gpu_ctx, oracle):
    """1920x1080 on a 128^3 volume: fused == hooks everywhere, and a 6-row oracle band"""
    sc = Scene(gpu_ctx, oracle, 128, 1920, 1080, tf=colour_tf(64, 1))
    outs = []
    for fused in (True, False):
        r = sc.renderer(fused=fused)
        r.slices = 32; r.numShadowSamples = 3
        r.reset(); r.render()
        outs.append((r.getTexture(), r.read(N.BUFFER_ACCUM), r.sample_count()))
        params = oracle_params(oracle, r)
        r.destroy()
    same_bits(outs[0][0].view(np.uint16), outs[1][0].view(np.uint16), "lao fused vs hooks render")
    same_bits(outs[0][1], outs[1][1], "lao fused vs hooks accumulation")
    assert outs[0][2] == outs[1][2]
    y0, y1 = 537, 543
    o = oracle.OracleRenderer('lao', sc.osc, sc.w, sc.h)
    o.lao = params
    fr = oracle.make_frame(sc.w, sc.h, sc.m, steps=32, extinction=100, y0=y0, y1=y1)
    o.reset(fr); o.render(fr)
    same_bits(outs[0][1][y0:y1], o.acc.reshape(sc.h, sc.w, 4)[y0:y1], "lao 1080p band")
    sc.gvol.destroy()
