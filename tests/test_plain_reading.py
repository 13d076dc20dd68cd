"""CPU: pins the CONTRACT oracle (oracle/vpt_oracle.c — what the HIP kernels are bit-exact against) to a PLAIN READING of
the reference's shaders (oracle/plain_reading.py: IEEE division / sqrt, libm transcendentals, natural summation order, no
fma, no reciprocal routines, nothing hoisted).  The reference holds no fixtures and its GLSL cannot run in the build
container ("parity unpinned" by the reference's own tests), so this is the strongest statement available about
"contract ~ GLSL": the two restatements were written separately, one from the kernels' side, one from the shader text.

Tolerances ("contract vs plain reading"):
  MIP   identical unorm8 frames except round-half ties of the quantisation: |d| <= 1 level, <= 1 % of pixels
  EAM   |d| <= 1 unorm8 level per channel (generate and the re-quantised running mean)
        (NEAREST filter: floor(s*N) is discontinuous, a last-bit difference of s*N on a voxel face picks the other voxel:
        <= 0.5 % of pixels may differ by more than one level)
  MCS   the first frame: >= 99.9 % of pixels within 1e-4 absolute per channel (a differing pixel = one tracking decision
        that fell the other way on a last-bit difference); same for the MCM photon after its first event
  MCS / MCM converged images: same seeds: mean |d| <= 1/4 of the Monte-Carlo noise between two runs with different seeds;
        image means per channel agree within K_SIGMA = 4 standard errors of that noise
"""
import numpy as np
import pytest

from conftest import default_matrix, orbit_camera
from oracle import plain_reading as P
from vpt_amd.scene import Transform, Node, mvp_inverse_matrix
from vpt_amd.synthetic import sphere_volume, ramp_tf, colour_tf

K_SIGMA = 4.0
W = H = 64
N = 32


def seed_k(k):
    return float(np.float32((k * 0.61803398875) % 1.0))


def scenes():
    vol = sphere_volume(N, noise=40.0)
    cam = mvp_inverse_matrix(orbit_camera(1.0, 0.5, -0.3, 1.9), Transform(Node()))
    return [
        ("default-camera default-tf linear", vol, "linear", None, default_matrix(1.0)),
        ("orbit ramp-tf linear", vol, "linear", ramp_tf(64), cam),
        ("orbit colour-tf nearest", vol, "nearest", colour_tf(32), cam),
    ]


@pytest.mark.parametrize("case", range(3))
def test_mip_contract_equals_plain_reading_up_to_quantisation_ties(oracle, case):
    O = oracle
    name, vol, filt, tf, m = scenes()[case]
    sc = O.OracleScene(vol, filt, tf=tf)
    ps = P.Scene(vol, filt, tf_rgba8=tf)
    # NEAREST: with offset 0 the sample at t = 1/2 of a ray through the cube centre sits exactly ON a voxel face (N even)
    for steps, offset in ((64, 0.0 if filt == "linear" else 0.13), (40, 0.37)):
        o = O.OracleRenderer("mip", sc, W, H)
        o.generate(O.make_frame(W, H, m, steps=steps, offset=offset))
        want = P.mip_generate(ps, W, H, m, np.float32(1.0 / steps), offset)
        got = o.frame.reshape(H, W)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        if filt == "nearest":                             # floor(s * N) may pick the neighbouring voxel on a last-bit difference
            assert (d > 1).mean() <= 0.005, (name, steps, float((d > 1).mean()))
        else:
            assert d.max() <= 1, (name, steps, int(d.max()))
        assert (d != 0).mean() <= 0.01, (name, steps, float((d != 0).mean()))
        assert (got > 0).mean() > 0.05                    # the scene is not empty


@pytest.mark.parametrize("case", range(3))
def test_eam_contract_within_one_level_of_plain_reading(oracle, case):
    O = oracle
    name, vol, filt, tf, m = scenes()[case]
    sc = O.OracleScene(vol, filt, tf=tf)
    ps = P.Scene(vol, filt, tf_rgba8=tf)
    o = O.OracleRenderer("eam", sc, W, H)
    o.reset(O.make_frame(W, H, m))
    acc = np.zeros((H, W, 4), np.uint8); acc[..., 3] = 255       # EAMRenderer.glsl:177-179
    for n, (ext, offset) in enumerate(((100.0, 0.0 if filt == "linear" else 0.13), (100.0, 0.61), (30.0, 0.23)), start=1):
        fr = O.make_frame(W, H, m, steps=64, offset=offset, extinction=ext, mix=1.0 / n)
        o.generate(fr); o.integrate(fr)
        frame = P.eam_generate(ps, W, H, m, np.float32(1.0 / 64), offset, ext)
        d = np.abs(o.frame.reshape(H, W, 4).astype(np.int32) - frame.astype(np.int32))
        if filt == "nearest":
            assert (d > 1).any(axis=-1).mean() <= 0.005, (name, n, float((d > 1).any(axis=-1).mean()))
        else:
            assert d.max() <= 1, (name, n, int(d.max()))
        assert (d != 0).mean() <= 0.02
        # the accumulator is re-quantised every frame: mix the CONTRACT's frame so that one-level differences do not compound
        acc = P.eam_integrate(acc, o.frame.reshape(H, W, 4), np.float32(1.0 / n))
        da = np.abs(o.acc.reshape(H, W, 4).astype(np.int32) - acc.astype(np.int32))
        assert da.max() <= 1, (name, n, int(da.max()))


def _mcs_pair(O, vol, filt, tf, m, ext, frames, seed0=1, light=(0.3, 0.5, 0.8)):
    l = np.asarray(light, np.float32); l = (l / np.sqrt((l * l).sum())).astype(np.float32)
    sc = O.OracleScene(vol, filt, tf=tf)
    ps = P.Scene(vol, filt, tf_rgba8=tf)
    o = O.OracleRenderer("mcs", sc, W, H)
    o.reset(O.make_frame(W, H, m))
    pacc = np.zeros((H, W, 4), np.float32); pacc[..., 3] = 1.0   # MCSRenderer.glsl:238-240
    first = None
    for n in range(1, frames + 1):
        s = seed_k(seed0 + n)
        fr = O.make_frame(W, H, m, seed=s, extinction=ext, light_dir=l, mix=1.0 / n)
        o.generate(fr); o.integrate(fr)
        pf = P.mcs_generate(ps, W, H, m, s, ext, l)
        if first is None:
            first = (o.frame.reshape(H, W, 4).copy(), pf)
        pacc = P.mcs_integrate(pacc, pf, np.float32(1.0 / n))
    return first, o.acc.reshape(H, W, 4).copy(), pacc


def test_mcs_first_frame_decisions_agree(oracle):
    for name, vol, filt, tf, m in scenes():
        (cf, pf), _, _ = _mcs_pair(oracle, vol, filt, tf, m, 8.0, 1)
        close = (np.abs(cf - pf) <= 1e-4).all(axis=-1)
        assert close.mean() >= 0.999, (name, float(close.mean()))
        assert np.abs(cf - pf)[close].max() <= 1e-4


def test_mcs_converged_images_agree(oracle):
    name, vol, filt, tf, m = scenes()[1]
    frames = 256
    _, ca, pa = _mcs_pair(oracle, vol, filt, tf, m, 8.0, frames)
    _, cb, _ = _mcs_pair(oracle, vol, filt, tf, m, 8.0, 64, seed0=5000)       # another stream: the Monte-Carlo noise scale
    noise = np.abs(ca[..., :3] - cb[..., :3]).mean()
    same = np.abs(ca[..., :3] - pa[..., :3]).mean()
    assert noise > 0
    assert same <= 0.25 * noise, (same, noise)
    se = (ca[..., :3] - cb[..., :3]).std(axis=(0, 1)) / np.sqrt(W * H)
    dm = np.abs(ca[..., :3].mean(axis=(0, 1)) - pa[..., :3].mean(axis=(0, 1)))
    assert (dm <= K_SIGMA * se + 1e-6).all(), (dm, se)


def _mcm_contract(O, vol, filt, tf, m, ext, g, passes, steps, seed0=1, max_bounces=8):
    sc = O.OracleScene(vol, filt, tf=tf)
    o = O.OracleRenderer("mcm", sc, W, H)
    o.reset(O.make_frame(W, H, m, seed=seed_k(seed0)))
    hist = []
    for k in range(1, passes + 1):
        before = (o.state[3].reshape(H, W, 4)[..., 3].copy(), o.state[1].reshape(H, W, 4)[..., 3].copy())
        o.integrate(O.make_frame(W, H, m, seed=seed_k(seed0 + k), extinction=ext, anisotropy=g, max_bounces=max_bounces, mcm_steps=steps))
        hist.append(before)
    return o, hist


def _mcm_plain(vol, filt, tf, m, ext, g, passes, steps, seed0=1, max_bounces=8, trace=None):
    ps = P.Scene(vol, filt, tf_rgba8=tf)
    st = P.mcm_reset(W, H, m, seed_k(seed0))
    for k in range(1, passes + 1):
        P.mcm_integrate(ps, st, m, seed_k(seed0 + k), ext, g, max_bounces, steps, trace=trace)
    return st


def test_mcm_reset_and_first_event_agree(oracle):
    O = oracle
    for (name, vol, filt, tf, m), g in zip(scenes(), (0.0, 0.6, -0.4)):
        o0, _ = _mcm_contract(O, vol, filt, tf, m, 6.0, g, 0, 1)
        st0 = P.mcm_reset(W, H, m, seed_k(1))
        pos = o0.state[0].reshape(H, W, 4); dr = o0.state[1].reshape(H, W, 4)
        # a ray that misses the cube is parked at from + tnear * dir with tnear up to ~1e5 (a slab it runs nearly
        # parallel to): those positions carry the relative error of tnear; near the cube the error is absolute
        near = (np.abs(st0.pos[0]) <= 4.0) & (np.abs(st0.pos[1]) <= 4.0) & (np.abs(st0.pos[2]) <= 4.0)
        onc = np.ones_like(near)                          # the photon starts ON the cube: the ray hits it
        for k in range(3):
            onc &= (st0.pos[k] >= -1e-3) & (st0.pos[k] <= 1.001)
        assert near.mean() > 0.5 and onc.mean() > 0.1
        for k in range(3):
            assert np.abs(pos[..., k] - st0.pos[k])[onc].max() <= 2e-5, name
            assert np.abs(pos[..., k] - st0.pos[k])[near].max() <= 2e-4, name
            assert np.allclose(dr[..., k], st0.dir[k], atol=2e-6, rtol=0), name
        dv = np.sqrt(sum((pos[..., k].astype(np.float64) - st0.pos[k]) ** 2 for k in range(3)))
        nv = np.sqrt(sum(st0.pos[k].astype(np.float64) ** 2 for k in range(3)))
        assert (dv <= 2e-4 + 2e-3 * nv).mean() >= 0.999, name   # relative to the length of the parked position (a ray nearly
                                                                 # parallel to the slab it is parked on is ill-conditioned: 1 pixel)
        assert (o0.state[3].reshape(H, W, 4) == np.array([1, 1, 1, 0], np.float32)).all()
        # one event: which branch did every pixel take?
        o1, hist = _mcm_contract(O, vol, filt, tf, m, 6.0, g, 1, 1)
        trace = []
        st1 = _mcm_plain(vol, filt, tf, m, 6.0, g, 1, 1, trace=trace)
        samples0, bounces0 = hist[0]
        fin_c = o1.state[3].reshape(H, W, 4)[..., 3] > samples0
        sct_c = ~fin_c & (o1.state[1].reshape(H, W, 4)[..., 3] > bounces0)
        code = trace[0]
        fin_p, sct_p = code >= 2, code == 1
        agree = (fin_c == fin_p) & (sct_c == sct_p)
        assert agree.mean() >= 0.999, (name, float(agree.mean()))
        assert sct_p.sum() > 0 and fin_p.sum() > 0 and (code == 0).sum() > 0, name     # every branch is exercised
        p1 = o1.state[0].reshape(H, W, 4); d1 = o1.state[1].reshape(H, W, 4)
        ok = agree.copy()
        n1 = np.sqrt(sum(st1.pos[k].astype(np.float64) ** 2 for k in range(3)))
        for k in range(3):
            ok &= np.abs(p1[..., k] - st1.pos[k]) <= 1e-4 + 2e-3 * np.maximum(n1 - 2.0, 0.0)   # absolute near the cube
            ok &= np.abs(d1[..., k] - st1.dir[k]) <= 1e-4
        assert ok.mean() >= 0.999, (name, float(ok.mean()))
        rad = o1.state[3].reshape(H, W, 4)
        for k in range(3):
            assert np.abs(rad[..., k] - st1.rad[k])[agree].max() <= 1e-5


def test_mcm_converged_images_agree(oracle):
    O = oracle
    name, vol, filt, tf, m = scenes()[1]
    passes, steps, ext, g = 96, 8, 6.0, 0.3
    oa, _ = _mcm_contract(O, vol, filt, tf, m, ext, g, passes, steps)
    ob, _ = _mcm_contract(O, vol, filt, tf, m, ext, g, 24, steps, seed0=7000)
    sp = _mcm_plain(vol, filt, tf, m, ext, g, passes, steps)
    ca = oa.state[3].reshape(H, W, 4)[..., :3]; cb = ob.state[3].reshape(H, W, 4)[..., :3]
    pa = np.stack(sp.rad, axis=-1)
    noise = np.abs(ca - cb).mean()
    same = np.abs(ca - pa).mean()
    assert noise > 0
    assert same <= 0.25 * noise, (same, noise)
    se = (ca - cb).std(axis=(0, 1)) / np.sqrt(W * H)
    dm = np.abs(ca.mean(axis=(0, 1)) - pa.mean(axis=(0, 1)))
    assert (dm <= K_SIGMA * se + 1e-6).all(), (dm, se)
    # the path counts are integers: most pixels completed exactly the same number of paths
    ns_c = oa.state[3].reshape(H, W, 4)[..., 3]
    assert (ns_c == sp.samples).mean() >= 0.97
    assert abs(float(ns_c.sum()) - float(sp.samples.sum())) <= 2e-3 * float(ns_c.sum())


# ---------------------------------------------------------------------------------------------------------------------------
# The BENCHMARK regime (BASELINE.json's headline configuration, scaled down in pixels and voxels only): aspect 16:9, the
# reference's default camera, its default 2x1 transfer function, 1x1 white environment, extinction 1, anisotropy 0, 8 bounces,
# 8 steps per pass.  What is special about it: ~80 % of the pixels never meet the cube (their photon is parked beside the cube at from + tnear * dir
# and leaves again at every event), and of the events of the pixels that do cross it most end outside.
# ---------------------------------------------------------------------------------------------------------------------------
BW, BH, BN = 128, 72, 32


def _bench_scene():
    return sphere_volume(BN, noise=48.0), default_matrix(BW / BH)


def _bench_contract(O, passes, seed0=1):
    vol, m = _bench_scene()
    o = O.OracleRenderer("mcm", O.OracleScene(vol, "linear"), BW, BH)
    o.reset(O.make_frame(BW, BH, m, seed=seed_k(seed0)))
    for k in range(1, passes + 1):
        o.integrate(O.make_frame(BW, BH, m, seed=seed_k(seed0 + k), extinction=1.0, anisotropy=0.0, max_bounces=8, mcm_steps=8))
    return o


def _bench_plain(passes, seed0=1, trace=None, steps=8):
    vol, m = _bench_scene()
    ps = P.Scene(vol, "linear")
    st = P.mcm_reset(BW, BH, m, seed_k(seed0))
    for k in range(1, passes + 1):
        P.mcm_integrate(ps, st, m, seed_k(seed0 + k), 1.0, 0.0, 8, steps, trace=trace)
    return st


def test_mcm_benchmark_regime_reset_first_event_and_parked_photons(oracle):
    O = oracle
    o0 = _bench_contract(O, 0)
    st0 = _bench_plain(0)
    pos = o0.state[0].reshape(BH, BW, 4); dr = o0.state[1].reshape(BH, BW, 4)
    onc = np.ones((BH, BW), bool)
    for k in range(3):
        onc &= (st0.pos[k] >= -1e-3) & (st0.pos[k] <= 1.001)
    assert 0.15 < onc.mean() < 0.30                        # the cube's silhouette: about a fifth of a 16:9 image
    far = ~onc
    nv = np.sqrt(sum(st0.pos[k].astype(np.float64) ** 2 for k in range(3)))
    assert nv[far].max() > 2.0                             # photons parked beside the cube on rays that run past it
    for k in range(3):
        assert np.abs(pos[..., k] - st0.pos[k])[onc].max() <= 2e-5
        assert np.allclose(dr[..., k], st0.dir[k], atol=2e-6, rtol=0)
    dv = np.sqrt(sum((pos[..., k].astype(np.float64) - st0.pos[k]) ** 2 for k in range(3)))
    assert (dv <= 2e-4 + 2e-3 * nv).mean() >= 0.999        # parked positions: relative to their length (tnear is ill-conditioned there)
    # one pass of ONE event: branches
    vol, m = _bench_scene()
    o1 = O.OracleRenderer("mcm", O.OracleScene(vol, "linear"), BW, BH)
    o1.reset(O.make_frame(BW, BH, m, seed=seed_k(1)))
    o1.integrate(O.make_frame(BW, BH, m, seed=seed_k(2), extinction=1.0, anisotropy=0.0, max_bounces=8, mcm_steps=1))
    trace = []
    st1 = _bench_plain(1, trace=trace, steps=1)
    fin_c = o1.state[3].reshape(BH, BW, 4)[..., 3] > 0
    sct_c = ~fin_c & (o1.state[1].reshape(BH, BW, 4)[..., 3] > 0)
    code = trace[0]
    agree = (fin_c == (code >= 2)) & (sct_c == (code == 1))
    assert agree.mean() >= 0.999, float(agree.mean())
    assert (code == 3)[far].all() and fin_c[far].all()     # every event of a pixel that misses the cube ends out of bounds
    assert (code == 3).mean() > 0.85                       # ... and so do most events overall
    rad = o1.state[3].reshape(BH, BW, 4)
    for k in range(3):
        assert np.abs(rad[..., k] - st1.rad[k])[agree].max() <= 1e-5
        # white 1x1 environment, transmittance 1: exactly 1 (the plain reading blends the one texel with itself: 1 - 2^-24 at worst)
        assert (rad[..., k][far] == 1.0).all() and np.abs(st1.rad[k][far] - 1.0).max() <= 2e-7


def test_mcm_benchmark_regime_converged(oracle):
    O = oracle
    passes = 64
    oa = _bench_contract(O, passes)
    ob = _bench_contract(O, 16, seed0=7000)
    sp = _bench_plain(passes)
    ca = oa.state[3].reshape(BH, BW, 4); cb = ob.state[3].reshape(BH, BW, 4)
    pa = np.stack(sp.rad, axis=-1)
    full = 8 * passes
    crossing = ca[..., 3] < full
    assert 0.15 < crossing.mean() < 0.35
    # pixels that miss the cube: one completed path per event, radiance exactly the environment's
    assert (sp.samples[~crossing] == full).all() and (ca[..., 3][~crossing] == full).all()
    assert np.abs(pa[~crossing] - 1.0).max() <= 2e-7 and (ca[..., :3][~crossing] == 1.0).all()
    # pixels that cross it
    same_n = (ca[..., 3] == sp.samples)[crossing].mean()
    assert same_n >= 0.97, float(same_n)
    d = np.abs(ca[..., :3] - pa).max(axis=-1)[crossing]
    noise = np.abs(ca[..., :3] - cb[..., :3]).max(axis=-1)[crossing].mean()
    assert noise > 0 and d.mean() <= 0.25 * noise, (float(d.mean()), float(noise))
    se = (ca[..., :3] - cb[..., :3])[crossing].std(axis=0) / np.sqrt(crossing.sum())
    dm = np.abs(ca[..., :3][crossing].mean(axis=0) - pa[crossing].mean(axis=0))
    assert (dm <= K_SIGMA * se + 1e-6).all(), (dm, se)
    # where the path counts agree the two restatements differ by rounding only
    eq = (ca[..., 3] == sp.samples) & crossing
    assert np.quantile(np.abs(ca[..., :3] - pa).max(axis=-1)[eq], 0.99) <= 1e-5
