"""GPU: the fast-arithmetic variant of the MCM integrate pass (VPT_OPTION_FAST_MATH — hardware rcp / rsq / sqrt / log / sin /
cos and algebraically equal shorter forms, vpt_kernels.h mcm_events_fast) against the CONTRACT oracle.

There is no bit-exact CPU twin of this variant; what is checked (tolerances as in tests/test_plain_reading.py):
  * the integer parts agree exactly: samples per pass = P * steps, PCG stream shared;
  * reset buffers are untouched by the option (bit-identical to the oracle);
  * after the first event >= 99.9 % of the pixels took the same branch as the oracle and sit at the same place (1e-4);
  * images converged with the SAME seeds differ by <= 1/4 of the Monte-Carlo noise between two different seed streams, image means
    per channel within K_SIGMA = 4 standard errors;
  * the option is refused by renderers that have no fast variant, and switching it off restores bit-exactness."""
import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.synthetic import colour_tf, ramp_tf, GoldenRatioRng

from conftest import orbit_camera
from test_gpu_parity import Scene, to_frame, assert_same_bits, MCM_BUFFERS, env_map

pytestmark = pytest.mark.gpu
K_SIGMA = 4.0


@pytest.mark.parametrize("g,env,filt", [(0.0, None, "linear"), (0.5, None, "linear"), (-0.4, (16, 8), "linear"), (0.3, None, "nearest")])
def test_fast_math_first_event_agrees_with_contract(gpu_ctx, oracle, g, env, filt):
    e = env_map(*env) if env else None
    sc = Scene(gpu_ctx, oracle, 40, 160, 96, filt, tf=colour_tf(256, 1), env=e, camera=orbit_camera(160 / 96))
    r = sc.renderer('mcm')
    r.set_option(N.OPTION_FAST_MATH, 1)
    r.extinction = 6.0; r.anisotropy = g; r.bounces = 4; r.steps = 1
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset()
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    for b, s in zip(MCM_BUFFERS, o.state):
        assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "reset buffer %d (the option does not touch the reset pass)" % b)
    n0 = o.state[3].reshape(sc.h, sc.w, 4)[..., 3].copy(); b0 = o.state[1].reshape(sc.h, sc.w, 4)[..., 3].copy()
    r.render()
    o.render(to_frame(oracle, sc, r._u))
    got = [r.read(b) for b in MCM_BUFFERS]
    want = [s.reshape(sc.h, sc.w, 4) for s in o.state]
    fin_w = want[3][..., 3] > n0; sct_w = ~fin_w & (want[1][..., 3] > b0)
    fin_g = got[3][..., 3] > n0; sct_g = ~fin_g & (got[1][..., 3] > b0)
    agree = (fin_w == fin_g) & (sct_w == sct_g)
    assert agree.mean() >= 0.999, float(agree.mean())
    assert fin_w.sum() > 0 and sct_w.sum() > 0 and (~fin_w & ~sct_w).sum() > 0          # every branch is exercised
    ok = agree.copy()
    far = np.sqrt((want[0][..., :3].astype(np.float64) ** 2).sum(axis=-1))
    for k in range(3):
        ok &= np.abs(got[0][..., k] - want[0][..., k]) <= 1e-4 + 2e-3 * np.maximum(far - 2.0, 0.0)
        ok &= np.abs(got[1][..., k] - want[1][..., k]) <= 1e-4
    assert ok.mean() >= 0.999, float(ok.mean())
    assert np.abs(got[3][..., :3] - want[3][..., :3])[agree].max() <= 1e-5
    assert np.abs(got[2][..., :3] - want[2][..., :3])[agree].max() <= 1e-5
    assert r.sample_count() == sc.w * sc.h
    r.destroy(); sc.gvol.destroy()


def _converge(sc, oracle, passes, fast, start=1, fused_play=False):
    r = sc.renderer('mcm', rng=GoldenRatioRng(start))
    r.set_option(N.OPTION_FAST_MATH, 1 if fast else 0)
    r.extinction = 5.0; r.anisotropy = 0.3; r.bounces = 6; r.steps = 8
    r.reset()
    if fused_play:
        r.play(passes, fused=True)
    else:
        for _ in range(passes):
            r.render()
    rad = r.read(N.BUFFER_MCM_RADIANCE).copy()
    n = r.sample_count()
    r.destroy()
    return rad, n


def test_fast_math_converged_image_within_noise(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 48, 192, 128, tf=ramp_tf(64), camera=orbit_camera(192 / 128, 0.5, -0.3, 1.9))
    passes = 160
    exact, n_e = _converge(sc, oracle, passes, False)
    fast, n_f = _converge(sc, oracle, passes, True)
    other, _ = _converge(sc, oracle, 40, False, start=9001)          # another seed stream: the Monte-Carlo noise scale
    assert n_e == n_f == sc.w * sc.h * 8 * passes
    ca, cf, cb = exact[..., :3], fast[..., :3], other[..., :3]
    noise = np.abs(ca - cb).mean(); same = np.abs(ca - cf).mean()
    assert noise > 0 and same <= 0.25 * noise, (same, noise)
    se = (ca - cb).std(axis=(0, 1)) / np.sqrt(sc.w * sc.h)
    dm = np.abs(ca.mean(axis=(0, 1)) - cf.mean(axis=(0, 1)))
    assert (dm <= K_SIGMA * se + 1e-6).all(), (dm, se)
    # path counts: integers; nearly every pixel completed the same number of paths
    assert (exact[..., 3] == fast[..., 3]).mean() >= 0.97
    assert abs(float(exact[..., 3].sum()) - float(fast[..., 3].sum())) <= 2e-3 * float(exact[..., 3].sum())
    # the fused-pass launch (VPT_PLAY_FUSED) runs the same fast events: identical to pass-by-pass launches of the fast variant
    fast2, _ = _converge(sc, oracle, 24, True, fused_play=True)
    fast3, _ = _converge(sc, oracle, 24, True)
    assert_same_bits(fast2, fast3, "fast variant: fused passes vs single passes")
    sc.gvol.destroy()


def test_fast_math_option_scope(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 24, 64, 48, tf=colour_tf(64, 1))
    for kind in ("mip", "eam", "mcs"):
        r = sc.renderer(kind)
        with pytest.raises(Exception):
            r.set_option(N.OPTION_FAST_MATH, 1)
        r.destroy()
    # on, then off again: bit-exact against the oracle as before
    r = sc.renderer('mcm')
    r.set_option(N.OPTION_FAST_MATH, 1); r.set_option(N.OPTION_FAST_MATH, 0)
    r.extinction = 7.0; r.steps = 5
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    for _ in range(3):
        r.render(); o.render(to_frame(oracle, sc, r._u))
    for b, s in zip(MCM_BUFFERS, o.state):
        assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "state buffer %d with the option switched off" % b)
    r.destroy(); sc.gvol.destroy()


# ---------------------------------------------------------------------------------------------------------------------------
# ONE hop: the HIP fast-arithmetic variant against the PLAIN READING of the shaders (oracle/plain_reading.py: IEEE / libm, natural
# order, nothing hoisted) — not via the contract oracle — in the benchmark regime: aspect 16:9, default camera, default 2x1 transfer
# function, white environment, extinction 1, anisotropy 0, 8 bounces, 8 steps (tests/test_plain_reading.py holds the contract
# oracle to the same reading in the same regime on the CPU).
# ---------------------------------------------------------------------------------------------------------------------------
def _regime(gpu_ctx, oracle, w=128, h=72, n=32):
    from vpt_amd.scene import default_camera
    return Scene(gpu_ctx, oracle, n, w, h, camera=default_camera(w / h), noise=48.0)


def _plain_run(sc, passes, steps=8, trace=None, start=0):
    from oracle import plain_reading as P
    rng = GoldenRatioRng(start)
    ps = P.Scene(sc.vol, "linear")
    st = P.mcm_reset(sc.w, sc.h, sc.m, float(np.float32(rng())))
    for _ in range(passes):
        P.mcm_integrate(ps, st, sc.m, float(np.float32(rng())), 1.0, 0.0, 8, steps, trace=trace)
    return st


@pytest.mark.parametrize("classes", [1, 0])
def test_fast_math_against_the_plain_reading_in_the_benchmark_regime(gpu_ctx, oracle, classes):
    sc = _regime(gpu_ctx, oracle)
    w, h = sc.w, sc.h

    def gpu(passes, steps=8, start=0):
        r = sc.renderer('mcm', rng=GoldenRatioRng(start))
        r.set_option(N.OPTION_FAST_MATH, 1)
        r.set_option(N.OPTION_TILE_CLASSES, classes)
        r.set_option(N.OPTION_SPLIT_STREAMS, 2)
        r.steps = steps
        r.reset()
        st0 = [r.read(b).copy() for b in MCM_BUFFERS]
        for _ in range(passes):
            r.render()
        out = [r.read(b).copy() for b in MCM_BUFFERS]
        assert r.sample_count() == w * h * steps * passes
        r.destroy()
        return st0, out

    # reset state and the first event
    st0, st1 = gpu(1, steps=1)
    p0 = _plain_run(sc, 0)
    onc = np.ones((h, w), bool)
    for k in range(3):
        onc &= (p0.pos[k] >= -1e-3) & (p0.pos[k] <= 1.001)
    nv = np.sqrt(sum(p0.pos[k].astype(np.float64) ** 2 for k in range(3)))
    for k in range(3):
        assert np.abs(st0[0][..., k] - p0.pos[k])[onc].max() <= 2e-5
        assert np.allclose(st0[1][..., k], p0.dir[k], atol=2e-6, rtol=0)
    dv = np.sqrt(sum((st0[0][..., k].astype(np.float64) - p0.pos[k]) ** 2 for k in range(3)))
    assert (dv <= 2e-4 + 2e-3 * nv).mean() >= 0.999
    trace = []
    p1 = _plain_run(sc, 1, steps=1, trace=trace)
    code = trace[0]
    fin_g = st1[3][..., 3] > 0
    sct_g = ~fin_g & (st1[1][..., 3] > 0)
    agree = (fin_g == (code >= 2)) & (sct_g == (code == 1))
    assert agree.mean() >= 0.999, float(agree.mean())
    ok = agree.copy()
    n1 = np.sqrt(sum(p1.pos[k].astype(np.float64) ** 2 for k in range(3)))
    for k in range(3):
        ok &= np.abs(st1[0][..., k] - p1.pos[k]) <= 1e-4 + 2e-3 * np.maximum(n1 - 2.0, 0.0)
        ok &= np.abs(st1[1][..., k] - p1.dir[k]) <= 1e-4
        assert np.abs(st1[3][..., k] - p1.rad[k])[agree].max() <= 1e-5
    assert ok.mean() >= 0.999, float(ok.mean())
    # converged with the same seeds
    passes = 64
    _, ga = gpu(passes)
    _, gb = gpu(16, start=7000)
    pa = _plain_run(sc, passes)
    rad_p = np.stack(pa.rad, axis=-1)
    full = 8 * passes
    crossing = pa.samples < full
    assert 0.15 < crossing.mean() < 0.35
    assert (ga[3][..., 3][~crossing] == full).all() and np.abs(ga[3][..., :3][~crossing] - rad_p[~crossing]).max() <= 2e-7
    assert (ga[3][..., 3] == pa.samples)[crossing].mean() >= 0.97
    d = np.abs(ga[3][..., :3] - rad_p).max(axis=-1)[crossing]
    noise = np.abs(ga[3][..., :3] - gb[3][..., :3]).max(axis=-1)[crossing].mean()
    assert noise > 0 and d.mean() <= 0.25 * noise, (float(d.mean()), float(noise))
    se = (ga[3][..., :3] - gb[3][..., :3])[crossing].std(axis=0) / np.sqrt(crossing.sum())
    dm = np.abs(ga[3][..., :3][crossing].mean(axis=0) - rad_p[crossing].mean(axis=0))
    assert (dm <= K_SIGMA * se + 1e-6).all(), (dm, se)
    eq = (ga[3][..., 3] == pa.samples) & crossing
    assert np.quantile(np.abs(ga[3][..., :3] - rad_p).max(axis=-1)[eq], 0.99) <= 1e-5
    sc.gvol.destroy()


# bounds of the H-size check below (per pixel: max over RGB of |radiance - oracle radiance|, after PASSES passes; one path whose fate
# flips changes a running mean over n paths by <= 1/n, and the rest of that pass's events with it)
H_PASSES = 256
# measured (gpurun_out/r04/fast_math_band.log): see H_MEASURED; the bounds are <= 10 x the measured figures (round 3's were 50-100 x)
H_MEASURED = {"equal_counts_crossing": 0.99893, "mean_abs_d_crossing": 1.19e-6, "p999_abs_d_crossing": 4.5e-5, "max_abs_d_crossing": 2.26e-3}
H_BOUNDS = {"min_equal_counts_crossing": 0.995, "mean_abs_d_crossing": 1.2e-5, "p999_abs_d_crossing": 4.5e-4, "max_abs_d_crossing": 2.3e-2}
# the same band as a viewer sees it — north_star's "per-channel RGBA tolerance": the default Artistic tone mapper's RGBA8
# (ArtisticToneMapper.glsl:37-46; low 0, mid 0.5, high 1, saturation 1, gamma 2.2) of the fast variant against the oracle's tone-mapped frame
# measured: max 1 LSB, every channel value within 1 LSB, 0.0063 % of them different at all; bounds = 2 x measured (and the 99.9 % / 2 LSB the review asked for)
H_DISPLAY_BOUNDS = {"max_lsb": 2, "min_fraction_within_1_lsb": 0.999, "max_fraction_different": 1.3e-4}


def test_full_size_fast_math_two_streams_tile_classes_oracle_band(gpu_ctx, oracle):
    """the configuration bench.py's default line is quoted on — MCM 512^3 @ 1920x1080, fast-math, the library's defaults (tile classes, HIT |
    MISS kernels on two streams) — against the CONTRACT oracle on a 24-row band through the cube (same band as the bit-exact full-size test)
    after 256 passes, with the stated fast-math tolerance in radiance AND in displayed RGBA8 units; sample count = P * steps; the general
    kernel (classes off, one stream) gives the same bits"""
    sc = Scene(gpu_ctx, oracle, 512, 1920, 1080, noise=48.0)

    def run(classes, split):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_FAST_MATH, 1)
        if not classes:                                        # (the defaults ARE classes on two streams: only the comparison run sets options)
            r.set_option(N.OPTION_TILE_CLASSES, 0)
            r.set_option(N.OPTION_SPLIT_STREAMS, split)
        r.set_option(N.OPTION_VERIFY_TILE_CLASSES, 1)
        r.reset()
        for _ in range(H_PASSES):
            r.render()
        tm = vpt_amd.ToneMapperFactory('artistic')(gpu_ctx, r, {'resolution': (sc.w, sc.h)})
        tm.render()
        out = (r.read(N.BUFFER_MCM_RADIANCE).copy(), r.getTexture().copy(), r.sample_count(), r.tile_classes(), tm.getTexture().copy())
        tm.destroy(); r.destroy()
        return out

    rad, img, ns, cls, shown = run(1, 2)
    assert ns == sc.w * sc.h * 8 * H_PASSES
    assert cls[1] > 0.7 * (cls[0] + cls[1]) and cls[2] == 0
    rad0, img0, _, _, shown0 = run(0, 1)
    assert_same_bits(rad, rad0, "tile classes on two streams vs the general kernel on one"); assert_same_bits(img, img0, "render buffer")
    assert_same_bits(shown, shown0, "tone-mapped frame")
    y0, y1 = 528, 552
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    rng = GoldenRatioRng()
    fr = oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(rng()), y0=y0, y1=y1, nthreads=16)
    o.reset(fr)
    for _ in range(H_PASSES):
        fr.seed = float(np.float32(rng()))
        o.integrate(fr)
    want = o.state[3].reshape(sc.h, sc.w, 4)[y0:y1]
    got = rad[y0:y1]
    full = 8.0 * H_PASSES
    crossing = want[..., 3] < full
    assert 0.25 < crossing.mean() < 0.45                      # the band runs through the cube's silhouette (660 of 1920 columns)
    assert (got[..., 3] == want[..., 3])[~crossing].all() and np.abs(got[..., :3] - want[..., :3])[~crossing].max() <= 1e-6
    d = np.abs(got[..., :3].astype(np.float64) - want[..., :3]).max(axis=-1)[crossing]
    eqc = float((got[..., 3] == want[..., 3])[crossing].mean())
    stats = (eqc, float(d.mean()), float(np.quantile(d, 0.999)), float(d.max()))
    print("H-size fast-math band after %d passes: equal counts %.5f, |d| mean %.3e p99.9 %.3e max %.3e" % ((H_PASSES,) + stats))
    # displayed units: the oracle's render pass (RGBA16F) through the oracle's Artistic tone mapper against the library's tone-mapped frame
    o.render_frame(fr)
    want8 = oracle.tonemap('artistic', o.image_f16()[y0:y1]).astype(np.int32)
    got8 = shown[y0:y1].astype(np.int32)
    lsb = np.abs(got8 - want8)[crossing]                      # [pixels crossing][4 channels]
    assert (got8 == want8)[~crossing].all()
    disp = (int(lsb.max()), float((lsb <= 1).mean()), float((lsb != 0).mean()))
    print("H-size fast-math band, Artistic RGBA8: max %d LSB, within 1 LSB %.6f of the channel values, different at all %.6f" % disp)
    assert eqc >= H_BOUNDS["min_equal_counts_crossing"], stats
    assert d.mean() <= H_BOUNDS["mean_abs_d_crossing"] and np.quantile(d, 0.999) <= H_BOUNDS["p999_abs_d_crossing"] and d.max() <= H_BOUNDS["max_abs_d_crossing"], stats
    assert disp[0] <= H_DISPLAY_BOUNDS["max_lsb"] and disp[1] >= H_DISPLAY_BOUNDS["min_fraction_within_1_lsb"] and disp[2] <= H_DISPLAY_BOUNDS["max_fraction_different"], disp
    sc.gvol.destroy()
