"""CPU: the tone-map oracle (oracle/vpt_tonemap_oracle.c) against float64 evaluations of the reference's formulas
(src/glsl/tonemappers/*.glsl, written out again here in numpy), its exp / pow / half routines against numpy, and the
committed golden fixture.  The reference has no tests of its own for this path: parity is pinned by these closed forms."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_exp_accuracy_and_specials(oracle):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-87.0, 88.0, 60001), np.random.default_rng(1).uniform(-10, 10, 40000)]).astype(np.float32)
    got = np.array([L.vpo_expf(float(x)) for x in xs], dtype=np.float64)
    want = np.exp(xs.astype(np.float64))
    ulp = np.abs(got - want) / np.spacing(want.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 2.0, ulp.max()
    assert L.vpo_expf(0.0) == 1.0
    assert L.vpo_expf(float('-inf')) == 0.0 and L.vpo_expf(-200.0) == 0.0
    assert L.vpo_expf(float('inf')) == float('inf') and L.vpo_expf(100.0) == float('inf')
    assert np.isnan(L.vpo_expf(float('nan')))
    # gradual underflow is kept: exp(-100) is a subnormal float
    assert 0.0 < L.vpo_expf(-100.0) < 1.2e-38


def test_pow_accuracy_and_specials(oracle):
    L = oracle.lib()
    r = np.random.default_rng(2)
    x = r.uniform(1e-4, 60.0, 50000).astype(np.float32); y = r.uniform(0.05, 3.0, 50000).astype(np.float32)
    got = np.array([L.vpo_powf(float(a), float(b)) for a, b in zip(x, y)], dtype=np.float64)
    want = np.power(x.astype(np.float64), y.astype(np.float64))
    assert (np.abs(got - want) / want).max() < 4e-6           # y*log(x) amplifies log's 2 ulp; GLSL's pow = exp2(y*log2(x)) behaves alike
    assert L.vpo_powf(1.0, 1 / 2.2) == 1.0                    # alpha of the curve mappers stays exactly 1
    assert L.vpo_powf(0.0, 0.45) == 0.0
    assert np.isnan(L.vpo_powf(-1.0, 0.45))                   # GLSL leaves x < 0 undefined: NaN here, 0 after the unorm8 write
    assert L.vpo_powf(float('inf'), 0.5) == float('inf')


def test_half_to_float_exhaustive(oracle):
    L = oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    got = np.array([L.vpo_f16_to_f32(int(b)) for b in bits], dtype=np.float32)
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all()
    assert (got[~nan].view(np.uint32) == want[~nan].view(np.uint32)).all()


# ---- float64 restatement of the shaders (independent of the C code) -----------------------------------------------
def _curve64(kind, x):
    if kind == "reinhard":
        return x / (1 + x)
    if kind == "reinhard2":
        return (x * (1 + x / 16.0)) / (1 + x)
    if kind == "uncharted2":
        def t(v):
            A, B, C, D, E, F = 0.15, 0.50, 0.10, 0.20, 0.02, 0.30
            return ((v * (A * v + C * B) + D * E) / (v * (A * v + B) + D * F)) - E / F
        return t(2.0 * x) * (1.0 / t(11.2))
    if kind == "filmic":
        X = np.maximum(0.0, x - 0.004)
        return ((X * (6.2 * X + 0.5)) / (X * (6.2 * X + 1.7) + 0.06)) ** 2.2
    if kind == "unreal":
        return x / (x + 0.155) * 1.019
    if kind == "aces":
        return np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1)
    if kind == "lottes":
        a, d, hdr, mi, mo = 1.6, 0.977, 8.0, 0.18, 0.267
        b = (-mi ** a + hdr ** a * mo) / ((hdr ** (a * d) - mi ** (a * d)) * mo)
        c = (hdr ** (a * d) * mi ** a - hdr ** a * mi ** (a * d) * mo) / ((hdr ** (a * d) - mi ** (a * d)) * mo)
        return x ** a / (x ** (a * d) * b + c)
    if kind == "uchimura":
        P, a, m, l, c, b = 1.0, 1.0, 0.22, 0.4, 1.33, 0.0
        l0 = ((P - m) * l) / a
        S0, S1 = m + l0, m + a * l0
        C2 = (a * P) / (P - S1); CP = -C2 / P
        t = np.clip(x / m, 0, 1)
        w0 = 1 - t * t * (3 - 2 * t)
        w2 = (x >= m + l0).astype(np.float64)
        w1 = 1 - w0 - w2
        T = m * (x / m) ** c + b
        S = P - (P - S1) * np.exp(CP * (x - S0))
        Lin = m + a * (x - m)
        return T * w0 + Lin * w1 + S * w2
    raise KeyError(kind)


def _tonemap64(kind, img, p):
    c = img.astype(np.float64)
    np.seterr(invalid='ignore')
    if kind == "artistic":
        v = (c - p["low"]) / (p["high"] - p["low"])
        gray = 1 / np.sqrt(3.0)
        d = v[..., :3].sum(-1, keepdims=True) * gray
        rgb = d * gray * (1 - p["saturation"]) + v[..., :3] * p["saturation"]
        e = (-np.log((p["mid"] - p["low"]) / (p["high"] - p["low"])) / np.log(2.0)) / p["gamma"]
        out = np.concatenate([rgb ** e, np.ones_like(d)], -1)
    elif kind == "range":
        out = ((c - p["min"]) / (p["max"] - p["min"])) ** (1 / p["gamma"])
    else:
        out = np.concatenate([_curve64(kind, c[..., :3] * p["exposure"]) ** (1 / p["gamma"]), np.ones_like(c[..., :1])], -1)
    with np.errstate(invalid='ignore'):
        out = np.where(np.isnan(out), 0.0, out)               # pow of a negative base: undefined in GLSL, 0 on the target by contract
    return np.clip(out, 0, 1) * 255.0


DEFAULTS = dict(low=0.0, mid=0.5, high=1.0, saturation=1.0, min=0.0, max=1.0, exposure=1.0, gamma=2.2)
VARIANTS = {"artistic": dict(low=0.05, mid=0.35, high=3.0, saturation=0.6, gamma=1.9), "range": dict(min=0.01, max=5.0, gamma=1.4)}


@pytest.mark.parametrize("kind", ["artistic", "range", "reinhard", "reinhard2", "uncharted2", "filmic", "unreal", "aces", "lottes", "uchimura"])
def test_oracle_matches_float64_formulas(oracle, kind):
    rng = np.random.default_rng(11)
    img = np.ones((4000, 4), dtype=np.float32)
    img[:, :3] = rng.uniform(0.02, 6.0, size=(4000, 3)) ** 2 / 5.0          # positive HDR values (x < 0 is undefined in GLSL)
    img = img.astype(np.float16)
    for params in ({}, VARIANTS.get(kind, dict(exposure=1.7, gamma=1.8))):
        p = dict(DEFAULTS); p.update(params)
        got = oracle.tonemap(kind, img, **params).astype(np.float64)
        want = _tonemap64(kind, img.astype(np.float32), p)
        # the oracle rounds to nearest: it may differ from round(float64) only where the exact value sits on a rounding boundary
        assert np.abs(got - want).max() <= 0.5 + 2e-3, (kind, params, np.abs(got - want).max())


def test_range_identity_is_the_unorm8_write(oracle):
    """Range with min 0, max 1, gamma 1 is pow(x, 1) = exp(log x): within 1 LSB of the plain unorm8 write, exact at 0 and 1"""
    vals = np.linspace(0, 1, 1001, dtype=np.float32).astype(np.float16)
    img = np.stack([vals, vals, vals, np.ones_like(vals)], -1)
    got = oracle.tonemap("range", img, gamma=1.0)
    want = np.rint(vals.astype(np.float32) * 255.0)
    assert np.abs(got[:, 0].astype(np.float32) - want).max() <= 1
    assert (got[0] == [0, 0, 0, 255]).all() and (got[-1] == [255, 255, 255, 255]).all()


def test_undefined_inputs_are_defined_here(oracle):
    img = np.array([[-1.0, np.nan, np.inf, 1.0], [0.0, -0.0, 65504.0, 1.0]], dtype=np.float16)
    out = oracle.tonemap("reinhard", img)
    assert out[0, 0] == 0 and out[0, 1] == 0        # pow of a negative / NaN -> NaN -> 0
    assert out[0, 2] == 0                           # inf/(1+inf) = NaN -> 0
    assert (out[1] == [0, 0, 255, 255]).all()


def test_golden_fixture(oracle):
    fx = json.load(open(os.path.join(HERE, "golden", "tonemap_r01.json")))
    src = np.array(fx["source_rgba16f_bits"], dtype=np.uint16).view(np.float16).reshape(-1, 4)
    assert len(fx["cases"]) == 20
    for case in fx["cases"]:
        got = oracle.tonemap(case["kind"], src, **case["params"])
        assert got.reshape(-1).tolist() == case["rgba8"], case["kind"]


def test_factory_names_follow_the_reference():
    import vpt_amd
    names = ['artistic', 'range', 'reinhard', 'reinhard2', 'uncharted2', 'filmic', 'unreal', 'aces', 'lottes', 'uchimura']   # ToneMapperFactory.js:14-23
    for i, n in enumerate(names):
        assert vpt_amd.ToneMapperFactory(n)._KIND == i
    with pytest.raises(RuntimeError, match='No suitable class'):
        vpt_amd.ToneMapperFactory('linear')
