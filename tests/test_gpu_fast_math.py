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
