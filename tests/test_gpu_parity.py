"""GPU parity: the HIP path (through the C-ABI, via the host mirror) against the CPU oracle on the same
seeded inputs.  Bar: BIT-EXACT for every buffer (u8, f32 bit patterns, f16 bit patterns) — the kernels and
the oracle implement the same fp32 operation sequence."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import Transform, Node, default_camera, mvp_inverse_matrix
from vpt_amd.synthetic import sphere_volume, colour_tf, ramp_tf, GoldenRatioRng

from conftest import orbit_camera

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32, 1: np.uint8}[a.dtype.itemsize])


def assert_same_bits(got, want, what):
    g, w = bits(got).reshape(-1), bits(want).reshape(-1)
    assert g.shape == w.shape, what
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, "%s: %d of %d elements differ, first at %d: got %r want %r" % (
        what, bad.size, g.size, bad[0], np.asarray(got).reshape(-1)[bad[0]], np.asarray(want).reshape(-1)[bad[0]])


# ---------------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------------
def test_math_probes_bit_exact(gpu_ctx, oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    # log on the exact domain of random_uniform: k * 2^-32 (rounded), plus edge values
    k = rng.integers(1, 2 ** 32, size=200000, dtype=np.uint64)
    u = (k.astype(np.float32) * np.float32(2.0 ** -32)).astype(np.float32)
    u = np.concatenate([u, np.array([0.0, 1.0, 2.0 ** -32, 0.5, 0.70710678, 0.70710677, 1e-30, 3.0, np.inf], np.float32)])
    got = gpu_ctx.probe_math(N.PROBE_LOG, u)
    want = np.array([L.vpo_logf(float(x)) for x in u], dtype=np.float32)
    assert_same_bits(got, want, "log")
    # sin / cos on [0, 2*pi]
    a = (rng.random(100000, dtype=np.float32) * np.float32(6.28318530718)).astype(np.float32)
    a = np.concatenate([a, np.array([0.0, 6.28318530718, 1.5707964, 3.1415927, 4.712389], np.float32)])
    s, c = C.c_float(), C.c_float()
    ws, wc = np.empty_like(a), np.empty_like(a)
    for i, x in enumerate(a):
        L.vpo_sincosf(float(x), C.byref(s), C.byref(c)); ws[i] = s.value; wc[i] = c.value
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_SIN, a), ws, "sin")
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_COS, a), wc, "cos")
    # asin / atan2
    x = np.concatenate([rng.uniform(-1, 1, 50000).astype(np.float32), np.array([-1, 1, 0, 0.5, -0.5, 1.0000001], np.float32)])
    want = np.array([L.vpo_asinf(float(v)) for v in x], dtype=np.float32)
    got = gpu_ctx.probe_math(N.PROBE_ASIN, x)
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all()
    assert_same_bits(got[~nan], want[~nan], "asin")
    yx = rng.normal(size=(50000, 2)).astype(np.float32)
    yx = np.concatenate([yx, np.array([[0, 0], [0, -1], [1, 0], [-1, 0], [0, 1], [-0.0, -1], [3, 3], [-2, 2]], np.float32)])
    want = np.array([L.vpo_atan2f(float(p[0]), float(p[1])) for p in yx], dtype=np.float32)
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_ATAN2, yx.reshape(-1)), want, "atan2")


def test_rcp_rsqrt_minmax_probes_bit_exact(gpu_ctx, oracle):
    """the contract's software reciprocal / rsqrt / sqrt and the minNum / maxNum semantic of v_min_f32 / v_max_f32"""
    L = oracle.lib()
    rng = np.random.default_rng(4)
    x = (np.exp(rng.uniform(-85, 85, 100000)) * rng.choice([-1.0, 1.0], 100000)).astype(np.float32)
    x = np.concatenate([x, np.float32([1, -1, 2, 0.5, 3, 1e-38, -1e-38, 3e38, 255, 1920, 1080])])
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_RCP, x), np.array([L.vpo_rcp_nr(float(v)) for v in x], np.float32), "rcp_nr")
    xz = np.concatenate([x, np.float32([0.0, -0.0, 1e-40, -1e-40, 1e-45])])
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_RCPZ, xz), np.array([L.vpo_rcp_nrz(float(v)) for v in xz], np.float32), "rcp_nrz")
    xp = np.concatenate([np.abs(x), np.float32([0.0, 1.0, 4.0, 1e-30])])
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_SQRT, xp), np.array([L.vpo_sqrt_nr(float(v)) for v in xp], np.float32), "sqrt_nr")
    xq = np.abs(x)
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_RSQRT, xq), np.array([L.vpo_rsqrt_nr(float(v)) for v in xq], np.float32), "rsqrt_nr")
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    pairs = np.float32([[1, 2], [2, 1], [0.0, -0.0], [-0.0, 0.0], [nan, 1], [1, nan], [-inf, inf], [inf, -inf], [nan, -0.0],
                        [-3, -3], [1e-45, 0], [-1e-45, 0]] + rng.normal(size=(5000, 2)).tolist())
    for probe, f in ((N.PROBE_MIN, L.vpo_min), (N.PROBE_MAX, L.vpo_max)):
        got = gpu_ctx.probe_math(probe, pairs.reshape(-1))
        want = np.array([f(float(a), float(b)) for a, b in pairs], np.float32)
        both_nan = np.isnan(want)
        assert (np.isnan(got) == both_nan).all()
        assert_same_bits(got[~both_nan], want[~both_nan], "min/max probe %d" % probe)
    k = rng.integers(0, 2 ** 32, size=100000, dtype=np.uint64)
    u = (k.astype(np.float32) * np.float32(2.0 ** -32)).astype(np.float32)
    u[:3] = [0.0, 1.0, 2.0 ** -32]
    assert_same_bits(gpu_ctx.probe_math(N.PROBE_LOG_UNIFORM, u), np.array([L.vpo_logf(float(v)) for v in u], np.float32), "log on uniforms")


def test_rng_and_half_probes(gpu_ctx, oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    st = rng.integers(0, 2 ** 32, size=100000, dtype=np.uint64).astype(np.uint32)
    st[:4] = [0, 1, 0xffffffff, 12345]
    got = gpu_ctx.probe_math(N.PROBE_PCG, st.view(np.float32)).view(np.uint32)
    want = np.array([L.vpo_pcg(int(s)) for s in st], dtype=np.uint32)
    assert (got == want).all()
    gotu = gpu_ctx.probe_math(N.PROBE_UNIFORM, st.view(np.float32))
    wantu = np.empty(st.size, np.float32)
    for i, s in enumerate(st):
        cs = C.c_uint32(int(s)); wantu[i] = L.vpo_random_uniform(C.byref(cs))
    assert_same_bits(gotu, wantu, "uniform")
    f = np.concatenate([rng.normal(size=100000).astype(np.float32) * np.float32(3),
                        np.array([0, -0.0, 1, 65504, 65519.9, 65520, 1e-8, 6e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 1e6, np.inf, -np.inf], np.float32),
                        (rng.random(20000, dtype=np.float32) * np.float32(1e-4)).astype(np.float32)])
    goth = gpu_ctx.probe_math(N.PROBE_F16, f).view(np.uint32).astype(np.uint16)
    wanth = np.array([L.vpo_f32_to_f16(float(v)) for v in f], dtype=np.uint16)
    assert (goth == wanth).all()
    assert (goth == f.astype(np.float16).view(np.uint16)).all()


# the last two: rows of 196 and 128 voxels take the LDS-staged strip re-layout for their first 3 / 1 strips and the generic one
# for the strip holding the last column; 68 = one whole strip plus 4 voxels
@pytest.mark.parametrize("dims", [(32, 32, 32), (17, 23, 9), (4, 4, 4), (1, 1, 1), (5, 64, 3), (7, 13, 196), (6, 5, 128), (3, 9, 68)])
@pytest.mark.parametrize("filt", ["linear", "nearest"])
def test_volume_sampler_and_transfer_function(gpu_ctx, oracle, dims, filt):
    """texture(uVolume,p) -> texture(uTransferFunction,(r,0)): bricked Z-order + LDS TF vs linear volume + 2D bilinear"""
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 256, size=dims, dtype=np.uint8)
    tf = colour_tf(37, 5)
    sc = oracle.OracleScene(vol, filt, tf=tf)
    v = vpt_amd.Volume.from_array(gpu_ctx, vol, filt)
    r = vpt_amd.MIPRenderer(gpu_ctx, v, default_camera(), None, {'resolution': 16})
    r.setTransferFunction(tf)
    p = rng.uniform(-0.3, 1.3, size=(20000, 3)).astype(np.float32)
    nx, ny, nz = dims[2], dims[1], dims[0]
    edge = np.array([[0, 0, 0], [1, 1, 1], [0.5 / nx, 0.5 / ny, 0.5 / nz], [1 - 0.5 / nx, 1 - 0.5 / ny, 1 - 0.5 / nz],
                     [np.inf, 0.5, 0.5], [-np.inf, 0.5, 0.5], [np.nan, 0.5, 0.5], [0.5, np.nan, -np.inf], [1e30, -1e30, 0.25]], np.float32)
    centres = np.stack([(rng.integers(0, nx, 500) + 0.5) / nx, (rng.integers(0, ny, 500) + 0.5) / ny,
                        (rng.integers(0, nz, 500) + 0.5) / nz], axis=1).astype(np.float32)
    p = np.concatenate([p, edge, centres])
    got = r.probe_sample(p)
    want = np.empty_like(got)
    L = oracle.lib()
    buf = np.empty(4, np.float32)
    for i in range(p.shape[0]):
        L.vpo_sample_volume_color(C.byref(sc.c), float(p[i, 0]), float(p[i, 1]), float(p[i, 2]), buf.ctypes.data_as(C.c_void_p))
        want[i] = buf
    assert_same_bits(got, want, "sampleVolumeColor %s %s" % (dims, filt))
    r.destroy(); v.destroy()


# ---------------------------------------------------------------------------------------------------
# renderer passes
# ---------------------------------------------------------------------------------------------------
class Scene:
    def __init__(self, gpu_ctx, oracle, n, w, h, filt="linear", tf=None, env=None, camera=None, noise=40.0, dims=None):
        self.vol = sphere_volume(n, noise=noise, dims=dims)
        self.w, self.h = w, h
        self.tf, self.env = tf, env
        self.osc = oracle.OracleScene(self.vol, filt, tf=tf, env=env)
        self.gvol = vpt_amd.Volume.from_array(gpu_ctx, self.vol, filt)
        self.camera = camera if camera is not None else default_camera(w / h)
        self.transform = Transform(Node())
        self.m = mvp_inverse_matrix(self.camera, self.transform)
        self.ctx = gpu_ctx

    def renderer(self, kind, **opts):
        o = {'resolution': (self.w, self.h), 'transform': self.transform, 'rng': GoldenRatioRng()}
        o.update(opts)
        r = vpt_amd.RendererFactory(kind)(self.ctx, self.gvol, self.camera, self.env, o)
        if self.tf is not None:
            r.setTransferFunction(self.tf)
        return r


def to_frame(oracle, sc, u, **kw):
    """oracle Frame from the uniforms the host mirror actually sent"""
    fr = oracle.make_frame(sc.w, sc.h, np.array(list(u.mvp_inverse), np.float32), **kw)
    fr.seed = u.rand_seed; fr.offset = u.offset; fr.step = u.step_size
    fr.extinction = u.extinction; fr.anisotropy = u.anisotropy
    fr.max_bounces = u.max_bounces; fr.steps = u.steps
    for i in range(3):
        fr.light_dir[i] = u.light_direction[i]
    fr.mix = u.mix; fr.blur = u.blur
    return fr


@pytest.mark.parametrize("filt", ["linear", "nearest"])
@pytest.mark.parametrize("fused", [False, True])
def test_mip_parity(gpu_ctx, oracle, filt, fused):
    sc = Scene(gpu_ctx, oracle, 64, 256, 256, filt, camera=orbit_camera(1.0))
    r = sc.renderer('mip', fused=fused)
    r.steps = 48                                   # non power of two: fp32 accumulation of 1/steps decides the trip count
    o = oracle.OracleRenderer('mip', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    for k in range(3):
        r.render()
        fr = to_frame(oracle, sc, r._u)
        o.render(fr)
        assert_same_bits(r.read(N.BUFFER_ACCUM), o.acc.reshape(sc.h, sc.w), "MIP accumulation frame %d" % k)
        if not fused:
            assert_same_bits(r.read(N.BUFFER_FRAME), o.frame.reshape(sc.h, sc.w), "MIP frame %d" % k)
        assert_same_bits(r.getTexture(), o.image_f16(), "MIP render frame %d" % k)
    assert r.sample_count() == o.samples
    assert o.acc.max() > 0 and o.acc.min() == 0
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("fused", [False, True])
def test_eam_parity(gpu_ctx, oracle, fused):
    sc = Scene(gpu_ctx, oracle, 48, 200, 120, tf=colour_tf(64, 1), camera=orbit_camera(200 / 120))
    r = sc.renderer('eam', fused=fused)
    r.slices = 40; r.extinction = 60
    o = oracle.OracleRenderer('eam', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    for k in range(4):
        r.render()
        fr = to_frame(oracle, sc, r._u)
        o.render(fr)
        assert_same_bits(r.read(N.BUFFER_ACCUM), o.acc.reshape(sc.h, sc.w, 4), "EAM accumulation frame %d" % k)
        if not fused:
            assert_same_bits(r.read(N.BUFFER_FRAME), o.frame.reshape(sc.h, sc.w, 4), "EAM frame %d" % k)
        assert_same_bits(r.getTexture(), o.image_f16(), "EAM render frame %d" % k)
    assert r.sample_count() == o.samples
    r.destroy(); sc.gvol.destroy()


def env_map(w, h, seed=5):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


@pytest.mark.parametrize("persistent", [1, 0])
@pytest.mark.parametrize("fused,env", [(False, None), (True, None), (True, (16, 8))])
def test_mcs_parity(gpu_ctx, oracle, fused, env, persistent):
    """persistent = 1: persistent waves with __ballot/__popcll active-ray compaction; 0: one thread per pixel"""
    e = env_map(*env) if env else None
    sc = Scene(gpu_ctx, oracle, 40, 160, 96, tf=colour_tf(256, 1), env=e, camera=orbit_camera(160 / 96))
    r = sc.renderer('mcs', fused=fused)
    r.set_option(N.OPTION_MCS_PERSISTENT, persistent)
    r.extinction = 12
    o = oracle.OracleRenderer('mcs', sc.osc, sc.w, sc.h)
    r.reset(); o.reset(oracle.make_frame(sc.w, sc.h, sc.m))
    for k in range(4):
        r.render()
        fr = to_frame(oracle, sc, r._u)
        o.render(fr)
        assert_same_bits(r.read(N.BUFFER_ACCUM), o.acc.reshape(sc.h, sc.w, 4), "MCS accumulation frame %d" % k)
        if not fused:
            assert_same_bits(r.read(N.BUFFER_FRAME), o.frame.reshape(sc.h, sc.w, 4), "MCS frame %d" % k)
        assert_same_bits(r.getTexture(), o.image_f16(), "MCS render frame %d" % k)
    assert r.sample_count() == o.samples
    r.destroy(); sc.gvol.destroy()


MCM_BUFFERS = [N.BUFFER_MCM_POSITION, N.BUFFER_MCM_DIRECTION, N.BUFFER_MCM_TRANSMITTANCE, N.BUFFER_MCM_RADIANCE]


@pytest.mark.parametrize("persistent", [0, 1, 2])
@pytest.mark.parametrize("fused,g,env,ext", [(False, 0.0, None, 1.0), (True, 0.0, None, 8.0), (True, 0.6, None, 8.0),
                                             (True, -0.4, (16, 8), 20.0)])
def test_mcm_parity(gpu_ctx, oracle, fused, g, env, ext, persistent):
    """persistent = 0: one workgroup per 16x16 tile (default); 1: persistent waves; 2: + next-segment state prefetch"""
    e = env_map(*env) if env else None
    sc = Scene(gpu_ctx, oracle, 40, 144, 80, tf=colour_tf(256, 1), env=e, camera=orbit_camera(144 / 80))
    r = sc.renderer('mcm', fused=fused)
    r.set_option(N.OPTION_MCM_PERSISTENT, persistent)
    r.extinction = ext; r.anisotropy = g; r.bounces = 3; r.steps = 6
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    r.reset()
    fr = oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()()))
    o.reset(fr)
    for b, s in zip(MCM_BUFFERS, o.state):
        assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "MCM reset buffer %d" % b)
    for k in range(5):
        r.render()
        fr = to_frame(oracle, sc, r._u)
        o.render(fr)
        for b, s in zip(MCM_BUFFERS, o.state):
            assert_same_bits(r.read(b), s.reshape(sc.h, sc.w, 4), "MCM state buffer %d pass %d" % (b, k))
        assert_same_bits(r.getTexture(), o.image_f16(), "MCM render pass %d" % k)
    assert r.sample_count() == o.samples == sc.w * sc.h * 6 * 5
    r.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("kind", ["mip", "eam", "mcs", "mcm"])
@pytest.mark.parametrize("world,rows", [(2, 8), (3, 5), (8, 8)])
def test_sharded_equals_unsharded(gpu_ctx, oracle, kind, world, rows):
    """image-plane sharding: every rank's rows, scattered back, equal the single-renderer image bit for bit"""
    sc = Scene(gpu_ctx, oracle, 32, 100, 70, tf=colour_tf(64, 1), camera=orbit_camera(100 / 70))

    def run(r):
        if kind in ('mcs', 'mcm'):
            r.extinction = 9
        r.reset()
        for _ in range(3):
            r.render()
        return r.getTexture()

    full = sc.renderer(kind)
    want = run(full)
    got = np.zeros_like(want)
    seen = np.zeros(sc.h, dtype=bool)
    for rank in range(world):
        r = sc.renderer(kind, shard=(rank, world, rows))
        img = run(r)
        rows_g = r.global_rows()
        assert img.shape[0] == rows_g.size
        for l, j in enumerate(rows_g):
            if j >= 0:
                assert not seen[j]
                seen[j] = True
                got[j] = img[l]
        r.destroy()
    assert seen.all()
    assert_same_bits(got, want, "%s sharded %d" % (kind, world))
    full.destroy(); sc.gvol.destroy()


def test_native_rccl_gather_single_rank(gpu_ctx, oracle):
    """vpt_gather_*: the RCCL pipeline below the C ABI with a one-rank communicator — frames equal the plain render()"""
    from vpt_amd.tiles import RcclFrameGather
    sc = Scene(gpu_ctx, oracle, 32, 100, 70, tf=colour_tf(64, 1), camera=orbit_camera(100 / 70))
    plain = sc.renderer('mcm'); plain.extinction = 9; plain.reset()
    shard = sc.renderer('mcm', shard=(0, 1, 8)); shard.extinction = 9; shard.reset()
    g = RcclFrameGather(shard, RcclFrameGather.unique_id(), 0, 1)
    for k in range(5):
        plain.render()
        g.render()
        if k in (0, 3, 4):
            assert_same_bits(g.frame(), plain.getTexture(), "gathered frame %d" % k)
    g.synchronize()
    assert_same_bits(shard.read(N.BUFFER_MCM_RADIANCE), plain.read(N.BUFFER_MCM_RADIANCE), "state after gather pipeline")
    # a captured hipGraph that holds the RCCL exchange stays refused (slower per frame and an abort after ~50
    # replays on ROCm 7.0 / RCCL 2.26): the call must fail with VPT_ERR_UNSUPPORTED and leave the pipeline usable
    import ctypes as C
    u, vars_ = shard._collect_frames(2)
    plain._collect_frames(2)                          # keep the two renderers' per-frame draws in step
    rc = N.lib().vpt_gather_play(g._h, C.byref(u), vars_.ctypes.data_as(C.c_void_p), 2, N.PLAY_GRAPH)
    assert rc == N.ERR_UNSUPPORTED
    g.render(); plain.render()
    assert_same_bits(g.frame(), plain.getTexture(), "gathered frame after the refused graph request")
    g.destroy(); plain.destroy(); shard.destroy(); sc.gvol.destroy()


def test_errors_are_loud(gpu_ctx):
    with pytest.raises(RuntimeError, match="No suitable class"):
        vpt_amd.RendererFactory('nope')
    r = vpt_amd.MIPRenderer(gpu_ctx, None, default_camera(), None, {'resolution': 32})
    r.reset()
    with pytest.raises(vpt_amd.VptError, match="no ready volume"):
        r.render()
    r.destroy()


@pytest.mark.parametrize("dims,cam", [((20, 24, 17), (0.6, -0.35, 1.7)), ((1, 9, 33), (2.2, 0.4, 1.2)), ((40, 40, 40), (0.3, 0.2, 0.35))])
def test_mcm_boundary_atlas_is_bit_identical_to_the_bricks(gpu_ctx, oracle, dims, cam):
    """VPT_OPTION_BOUNDARY_ATLAS: out-of-cube samples from the six face images vs from the bricks — every state buffer identical
    to each other and to the oracle (non-cubic volumes, a one-voxel-thick axis, a camera INSIDE the volume)"""
    sc = Scene(gpu_ctx, oracle, 24, 88, 56, tf=colour_tf(32, 1), camera=orbit_camera(88 / 56, *cam), dims=dims)
    outs = []
    for atlas in (1, 0):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_BOUNDARY_ATLAS, atlas)
        r.extinction = 5; r.steps = 7
        r.reset()
        us = []
        for _ in range(4):
            r.render(); us.append(r._u)
        outs.append([r.read(b) for b in MCM_BUFFERS] + [r.getTexture()])
        r.destroy()
    for x, y in zip(*outs):
        assert_same_bits(x, y, "atlas on vs off")
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    o.reset(oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(GoldenRatioRng()())))
    for u in us:
        o.render(to_frame(oracle, sc, u))
    for b, s_ in zip(outs[0][:4], o.state):
        assert_same_bits(b, s_.reshape(sc.h, sc.w, 4), "atlas path vs oracle")
    sc.gvol.destroy()


@pytest.mark.parametrize("fast", [0, 1])
def test_mcm_split_streams_give_identical_buffers(gpu_ctx, oracle, fast):
    """VPT_OPTION_SPLIT_STREAMS = 2, 3, 4: every pass as K tile-row ranges on K HIP streams — with reads, a reset, a transfer
    function change and a tone mapper in between (each joins the side streams), every buffer equals the one-stream run's"""
    sc = Scene(gpu_ctx, oracle, 32, 176, 150, tf=colour_tf(64, 1), camera=orbit_camera(176 / 150))

    def run(split):
        r = sc.renderer('mcm')
        r.set_option(N.OPTION_FAST_MATH, fast)
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
        r.extinction = 5; r.steps = 6
        r.reset()
        outs = []
        for k in range(7):
            r.render()
            if k in (0, 4):
                outs.append(r.read(N.BUFFER_MCM_RADIANCE).copy())      # a join in the middle of a sequence
        tm = vpt_amd.ToneMapperFactory('artistic')(gpu_ctx, r, {'resolution': (sc.w, sc.h)})
        tm.render(); outs.append(tm.getTexture().copy()); tm.destroy()
        r.setTransferFunction(ramp_tf(32))
        r.reset()                                                        # reset kernel on the context's stream, then split passes again
        for k in range(3):
            r.render()
        outs += [r.read(b).copy() for b in MCM_BUFFERS] + [r.getTexture().copy()]
        assert r.sample_count() == sc.w * sc.h * 6 * 10
        gpu_ctx.synchronize()
        r.destroy()
        return outs

    a = run(1)
    for split in (2, 3, 4):
        b = run(split)
        assert len(a) == len(b)
        for k, (x, y) in enumerate(zip(a, b)):
            assert_same_bits(y, x, "%d split streams, output %d" % (split, k))
    r = sc.renderer('mcm')
    with pytest.raises(vpt_amd.VptError):
        r.set_option(N.OPTION_SPLIT_STREAMS, 5)
    r.destroy()
    sc.gvol.destroy()


@pytest.mark.parametrize("kind", ["mip", "eam", "mcs", "iso", "depth", "lao"])
def test_split_streams_other_renderers(gpu_ctx, oracle, kind):
    """VPT_OPTION_SPLIT_STREAMS on the other sampling renderers (their passes are per-pixel too): fused render(), the three hooks
    as separate launches, eager and graph frame sequences (a captured sequence stays on the capturing stream) — accumulator and
    render buffer equal the one-stream run's; DOS refuses the option"""
    sc = Scene(gpu_ctx, oracle, 32, 176, 150, tf=colour_tf(64, 1), camera=orbit_camera(176 / 150))

    def run(split, fused):
        r = sc.renderer(kind, fused=fused)
        r.set_option(N.OPTION_SPLIT_STREAMS, split)
        r.reset()
        for _ in range(4):
            r.render()
        outs = [r.getTexture().copy()]
        if fused:
            r.play(3, use_graph=False)
            r.play(3, use_graph=True); r.play(3, use_graph=True)
            r.render()
            if kind != "lao":                        # LAO frames do not accumulate: fused sequences are refused for it
                # one launch running several passes, split over the streams: the ranges on the side streams take the sequence's
                # first frame index by value (PassArgs.frame_base), not from the device counter the context's stream advances
                r.play(3, fused=True); r.play(5, fused=True)
                r.render()
        outs += [r.read(N.BUFFER_ACCUM).copy(), r.getTexture().copy(), r.sample_count()]
        r.destroy()
        return outs

    for fused in (True, False):
        a, b = run(1, fused), run(3, fused)
        for k, (x, y) in enumerate(zip(a, b)):
            if isinstance(x, np.ndarray):
                assert_same_bits(y, x, "%s fused=%s, 3 streams vs 1, output %d" % (kind, fused, k))
            else:
                assert x == y
    sc.gvol.destroy()


@pytest.mark.parametrize("fast", [0, 1])
def test_mcm_play_frames_writes_every_frame(gpu_ctx, oracle, fast):
    """VPT_PLAY_FRAMES: one launch runs `count` passes with the photon state in registers AND writes every pass's frame into the
    frame ring — slot f equals the render buffer after the f-th of `count` render() calls, state and sample count equal too;
    mixed with plain render() calls, a sharded renderer, more frames than slots refused, other renderers refused"""
    sc = Scene(gpu_ctx, oracle, 32, 176, 150, tf=colour_tf(64, 1), camera=orbit_camera(176 / 150))
    for shard in (None, (1, 3, 8)):
        opts = {'shard': shard} if shard else {}
        a, b = sc.renderer('mcm', **opts), sc.renderer('mcm', **opts)
        for r in (a, b):
            r.set_option(N.OPTION_FAST_MATH, fast)
            r.extinction = 6
            r.reset()
        a.render(); b.render()
        for count in (5, 1, N.FRAME_SLOTS):
            want = []
            for _ in range(count):
                a.render(); want.append(a.getTexture().copy())
            b.play(count, frames=True)
            for f in range(count):
                assert_same_bits(b.read_frame_slot(f), want[f], "frame %d of %d (shard %r)" % (f, count, shard))
            assert_same_bits(b.getTexture(), want[-1], "render buffer after the sequence")
            with pytest.raises(vpt_amd.VptError):
                b.read_frame_slot(count)
        a.render(); b.render()
        for buf in MCM_BUFFERS:
            assert_same_bits(b.read(buf), a.read(buf), "state after frame sequences")
        assert a.sample_count() == b.sample_count()
        with pytest.raises(vpt_amd.VptError):
            b.play(N.FRAME_SLOTS + 1, frames=True)
        a.destroy(); b.destroy()
    m = sc.renderer('mip'); m.reset(); m.render()
    with pytest.raises(vpt_amd.VptError):
        m.play(2, frames=True)
    m.destroy()
    sc.gvol.destroy()


def test_split_streams_refused_for_dos(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 16, 64, 48)
    r = sc.renderer('dos')
    with pytest.raises(vpt_amd.VptError):
        r.set_option(N.OPTION_SPLIT_STREAMS, 3)
    r.destroy(); sc.gvol.destroy()


def test_mcm_split_streams_with_a_caller_owned_render_target():
    """A frame rendered into caller memory (vpt_renderer_set_render_target) is read by work the CALLER enqueues on the
    context's stream right behind render(): with VPT_OPTION_SPLIT_STREAMS on, such passes must not leave rows on a side
    stream.  Runs in a fresh process (tests/render_target_worker.py): the consumer is a torch copy on torch's stream, and a
    GPU-initialised torch stays out of this process."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "render_target_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and out.strip().splitlines()[-1] == "OK", out[-2000:]


@pytest.mark.parametrize("kind", ["mip", "eam", "mcs", "mcm"])
def test_wide_offset_tables_variant(gpu_ctx, oracle, kind):
    """the 64-bit brick-offset-table kernels (used above 4 GiB of bricked data, e.g. 2048^3) forced on a small volume"""
    sc = Scene(gpu_ctx, oracle, 24, 72, 40, tf=colour_tf(32, 1), camera=orbit_camera(72 / 40), dims=(20, 24, 17))
    outs = []
    for wide in (0, 1):
        sc.gvol.set_wide_tables(wide)
        r = sc.renderer(kind)
        if kind in ('mcs', 'mcm'):
            r.extinction = 7
        r.reset()
        for _ in range(3):
            r.render()
        outs.append((r.getTexture(), r.sample_count()))
        r.destroy()
    assert_same_bits(outs[0][0], outs[1][0], "%s wide vs narrow tables" % kind)
    assert outs[0][1] == outs[1][1]
    sc.gvol.destroy()


# ---------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: size-independent properties + an oracle band
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_scene(gpu_ctx, oracle):
    sc = Scene(gpu_ctx, oracle, 512, 1920, 1080, noise=48.0)       # the headline workload's volume and image plane
    yield sc
    sc.gvol.destroy()


def test_full_size_mcm_properties_and_oracle_band(gpu_ctx, oracle, full_scene):
    """MCM 512^3 @ 1920x1080 (headline config): fused == hook-by-hook, run-to-run determinism, 3-way row sharding ==
    unsharded, sample count == P*steps, and a 24-row band of the frame bit-identical to the CPU oracle."""
    sc = full_scene
    passes = 3

    def run(**opts):
        r = sc.renderer('mcm', **opts)
        r.reset()
        for _ in range(passes):
            r.render()
        out = (r.getTexture(), r.read(N.BUFFER_MCM_RADIANCE), r.sample_count(), r.global_rows())
        r.destroy()
        return out

    img, rad, ns, _ = run()
    assert ns == sc.w * sc.h * 8 * passes
    img2, rad2, _, _ = run()
    assert_same_bits(img2, img, "determinism"); assert_same_bits(rad2, rad, "determinism (radiance)")
    img3, rad3, _, _ = run(fused=False)
    assert_same_bits(img3, img, "fused vs hooks"); assert_same_bits(rad3, rad, "fused vs hooks (radiance)")
    got = np.zeros_like(img)
    for rank in range(3):
        simg, _, _, rows = run(shard=(rank, 3, 8))
        got[rows[rows >= 0]] = simg[rows >= 0]
    assert_same_bits(got, img, "3-way sharded vs unsharded")
    assert np.isfinite(img.astype(np.float32)).all() and (img[..., 3] == 1).all()
    # oracle on a band of rows through the middle of the cube (same seeds as the mirror drew)
    y0, y1 = 528, 552
    o = oracle.OracleRenderer('mcm', sc.osc, sc.w, sc.h)
    rng = GoldenRatioRng()
    fr = oracle.make_frame(sc.w, sc.h, sc.m, seed=np.float32(rng()), y0=y0, y1=y1, nthreads=8)
    o.reset(fr)
    for _ in range(passes):
        fr.seed = float(np.float32(rng()))
        o.integrate(fr)
    want = o.state[3].reshape(sc.h, sc.w, 4)[y0:y1]
    assert_same_bits(rad[y0:y1], want, "oracle band rows %d..%d" % (y0, y1))


@pytest.mark.parametrize("kind,n,props", [("eam", 256, {}), ("mcs", 512, {"extinction": 20}), ("mip", 256, {})])
def test_full_size_other_renderers_oracle_band(gpu_ctx, oracle, full_scene, kind, n, props):
    """C2 / C3-sized passes: fused == hooks and a 16-row oracle band at 1920x1080"""
    sc = full_scene if n == 512 else Scene(gpu_ctx, oracle, n, 1920, 1080, noise=0.0)
    frames = 2

    def run(fused):
        r = sc.renderer(kind, fused=fused)
        for k, v in props.items():
            setattr(r, k, v)
        r.reset()
        us = []
        for _ in range(frames):
            r.render()
            us.append(r._u)
        out = (r.getTexture(), r.read(N.BUFFER_ACCUM), r.sample_count(), us)
        r.destroy()
        return out

    img, acc, ns, us = run(True)
    img_h, acc_h, ns_h, _ = run(False)
    assert_same_bits(img_h, img, "%s fused vs hooks" % kind); assert_same_bits(acc_h, acc, "%s fused vs hooks (acc)" % kind)
    assert ns == ns_h and ns > 0
    y0, y1 = 532, 548
    o = oracle.OracleRenderer(kind, sc.osc, sc.w, sc.h)
    fr0 = oracle.make_frame(sc.w, sc.h, sc.m, y0=y0, y1=y1, nthreads=8)
    o.reset(fr0)
    for u in us:
        fr = to_frame(oracle, sc, u, y0=y0, y1=y1, nthreads=8)
        o.render(fr)
    shape = (sc.h, sc.w) if kind == "mip" else (sc.h, sc.w, 4)
    assert_same_bits(acc[y0:y1], o.acc.reshape(shape)[y0:y1], "%s oracle band" % kind)
    if sc is not full_scene:
        sc.gvol.destroy()


def test_gpu_matches_committed_contract_digests(gpu_ctx):
    """the HIP path against tests/golden/contract_r01.json alone (no oracle library involved): explicit uniforms through
    the C-ABI, SHA-256 of the buffers"""
    import importlib.util, json, os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_contract_fixture", os.path.join(gold, "make_contract_fixture.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(gold, "contract_r01.json")))["scenes"]
    sc_list, dm = mod.scenes()
    L = N.lib()
    seq = lambda k: float(np.float32((k * 0.61803398875) % 1.0))
    for sc in sc_list:
        vol = vpt_amd.Volume.from_array(gpu_ctx, sc["vol"], sc["filter"])
        cam = default_camera(sc["w"] / sc["h"])
        r = vpt_amd.RendererFactory(sc["kind"])(gpu_ctx, vol, cam, None, {'resolution': (sc["w"], sc["h"]), 'transform': Transform(Node())})
        if sc["tf"] is not None:
            r.setTransferFunction(sc["tf"])
        r._bind_volume()
        kw = sc["kw"]
        u = r._new_uniforms()
        u.rand_seed = seq(1); u.blur = 0.0
        u.step_size = float(np.float32(1.0 / kw.get("steps", 64)))
        u.extinction = float(np.float32(kw.get("extinction", 1.0))); u.anisotropy = float(np.float32(kw.get("anisotropy", 0.0)))
        u.max_bounces = kw.get("max_bounces", 8); u.steps = kw.get("mcm_steps", 8)
        for i, v in enumerate(kw.get("light_dir", (0.0, 0.0, 1.0))):
            u.light_direction[i] = float(np.float32(v))
        u.isovalue = float(np.float32(kw.get("isovalue", 0.5))); u.gradient_step = float(np.float32(kw.get("gradient_step", 0.005)))
        u.threshold = float(np.float32(kw.get("threshold", 0.1)))
        N.check(L.vpt_renderer_reset(r._h, C.byref(u)))
        for k in range(sc["frames"]):
            u.rand_seed = seq(k + 2); u.offset = seq(k + 2); u.mix = float(np.float32(1.0 / (k + 1)))
            if sc["kind"] == "dos":
                slices, samples = mod.dos_inputs(k)
                N.check(L.vpt_renderer_set_occlusion_samples(r._h, samples.ctypes.data_as(C.c_void_p), len(samples)))
                N.check(L.vpt_renderer_integrate_slices(r._h, C.byref(u), slices.ctypes.data_as(C.c_void_p), len(slices)))
                N.check(L.vpt_renderer_render_frame(r._h, None))
            else:
                N.check(L.vpt_renderer_render(r._h, C.byref(u)))
        bufs = [r.read(b) for b in MCM_BUFFERS] if sc["kind"] == "mcm" else \
               [r.read(N.BUFFER_ACCUM), r.read(N.BUFFER_DOS_OCCLUSION)] if sc["kind"] == "dos" else [r.read(N.BUFFER_ACCUM)]
        got = {"buffers": mod.digest(*bufs), "render_f16": mod.digest(r.getTexture().view(np.uint16)), "samples": r.sample_count()}
        assert got == want[sc["name"]], sc["name"]
        r.destroy(); vol.destroy()


@pytest.mark.parametrize("kind", ["mip", "eam", "mcs", "mcm", "depth", "iso"])
def test_play_frame_sequences_equal_render_calls(gpu_ctx, oracle, kind):
    """vpt_renderer_play: per-frame uniforms from a device table, eager and as a replayed hipGraph == N x render()"""
    sc = Scene(gpu_ctx, oracle, 32, 100, 70, tf=colour_tf(64, 1), camera=orbit_camera(100 / 70))

    def make(one_stream=False):
        r = sc.renderer(kind)
        if kind in ('mcs', 'mcm'):
            r.extinction = 9
        if one_stream:
            r.set_option(N.OPTION_SPLIT_STREAMS, 1); r.set_option(N.OPTION_TILE_CLASSES, 0)
        r.reset()
        return r

    buf = MCM_BUFFERS[3] if kind == "mcm" else N.BUFFER_ACCUM
    ref = make()
    for _ in range(10):
        ref.render()
    want_img, want_buf, want_ns = ref.getTexture(), ref.read(buf), ref.sample_count()
    eager = make()
    eager.play(3, use_graph=False); eager.play(7, use_graph=False)
    assert_same_bits(eager.getTexture(), want_img, "%s eager play" % kind); assert_same_bits(eager.read(buf), want_buf, "%s eager play buffer" % kind)
    assert eager.sample_count() == want_ns
    # VPT_PLAY_GRAPH replays a hipGraph where that is the faster form: a renderer on one stream without tile classes (the first case);
    # with the defaults the same call plays the sequence eagerly (the second) — identical buffers either way
    for one_stream in (True, False):
        graph = make(one_stream)
        graph.render(); graph.render()                      # warm: lazy allocations happen outside the capture
        graph.play(4, use_graph=True); graph.play(4, use_graph=True)     # second call replays the cached graph
        assert_same_bits(graph.getTexture(), want_img, "%s graph play" % kind); assert_same_bits(graph.read(buf), want_buf, "%s graph play buffer" % kind)
        assert graph.sample_count() == want_ns
        graph.destroy()
    graph = make()
    if kind == "mcm":
        fusedr = make()
        fusedr.play(4, fused=True); fusedr.play(1, fused=True); fusedr.render(); fusedr.play(4, fused=True)      # 10 passes, mixed with a plain render()
        assert_same_bits(fusedr.getTexture(), want_img, "mcm fused passes"); assert_same_bits(fusedr.read(buf), want_buf, "mcm fused passes buffer")
        for b in MCM_BUFFERS:
            assert_same_bits(fusedr.read(b), ref.read(b), "mcm fused passes state %d" % b)
        assert fusedr.sample_count() == want_ns
        fusedr.play(2, use_graph=True); ref.render(); ref.render()          # the graph path's device frame counter stayed in step
        assert_same_bits(fusedr.read(buf), ref.read(buf), "graph replay after fused passes")
        fusedr.destroy()
    else:
        fusedr = make()
        fusedr.play(4, fused=True); fusedr.play(1, fused=True); fusedr.render(); fusedr.play(4, fused=True)
        assert_same_bits(fusedr.getTexture(), want_img, "%s fused passes" % kind); assert_same_bits(fusedr.read(buf), want_buf, "%s fused passes buffer" % kind)
        assert fusedr.sample_count() == want_ns
        fusedr.destroy()
    for r in (ref, eager, graph):
        r.destroy()
    sc.gvol.destroy()


@pytest.mark.parametrize("split", [1, 3])
@pytest.mark.parametrize("root", [-1, 0])
def test_native_rccl_gather_play(gpu_ctx, oracle, root, split):
    """vpt_gather_play: frame sequences (kernel + cross-stream event edges + RCCL gather per frame) by one call (one-rank
    comm); root = -1 all_gather, root = 0 gather to the display rank (which renders in place into its receive slot);
    split = 3: every pass as three tile-row ranges on three streams that are never joined between frames — the communication
    stream waits for each range's event, and the ring's reverse edge (frame 16 on) holds all three streams back"""
    from vpt_amd.tiles import RcclFrameGather
    sc = Scene(gpu_ctx, oracle, 32, 100, 70, tf=colour_tf(64, 1), camera=orbit_camera(100 / 70))
    plain = sc.renderer('mcm'); plain.extinction = 9; plain.reset()
    shard = sc.renderer('mcm', shard=(0, 1, 8)); shard.extinction = 9
    shard.set_option(N.OPTION_SPLIT_STREAMS, split)
    shard.reset()
    g = RcclFrameGather(shard, RcclFrameGather.unique_id(), 0, 1, root=root)
    assert g.receives()
    g.render(); g.render()
    for _ in range(2):
        plain.render()
    for rep in range(3):
        g.play(6)
        for _ in range(6):
            plain.render()
        assert_same_bits(g.frame(), plain.getTexture(), "gathered frame after sequence %d" % rep)
    g.play(3); g.render()
    for _ in range(4):
        plain.render()
    assert_same_bits(g.frame(), plain.getTexture(), "odd-length sequence then single frame")
    g.play(5, fused=True)                                   # five passes in one launch, one gather
    for _ in range(5):
        plain.render()
    assert_same_bits(g.frame(), plain.getTexture(), "fused passes then one gather")
    assert_same_bits(shard.read(N.BUFFER_MCM_RADIANCE), plain.read(N.BUFFER_MCM_RADIANCE), "state")
    assert shard.sample_count() == plain.sample_count()
    shard.setResolution((64, 40))                           # the gather's buffers were sized for 100x70: refuse, do not overrun
    with pytest.raises(vpt_amd.VptError, match="re-create the gather"):
        g.render()
    g.destroy(); plain.destroy(); shard.destroy(); sc.gvol.destroy()


@pytest.mark.parametrize("world,rows", [(2, 8), (3, 5), (8, 8), (5, 16)])
def test_multi_rank_frame_assembly(gpu_ctx, oracle, world, rows):
    """what vpt_gather_read_frame does with a frame gathered from `world` ranks, emulated on one GPU: the ranks' send
    buffers (local rows incl. padding) stacked as the gather delivers them, re-ordered by k_assemble_rows == unsharded frame"""
    sc = Scene(gpu_ctx, oracle, 32, 100, 70, tf=colour_tf(64, 1), camera=orbit_camera(100 / 70))

    def run(**opts):
        r = sc.renderer('mcm', **opts)
        r.extinction = 9
        r.reset()
        for _ in range(2):
            r.render()
        out = r.getTexture()
        r.destroy()
        return out

    whole = run()
    parts = [run(shard=(rank, world, rows)) for rank in range(world)]
    assert len({p.shape for p in parts}) == 1                      # every rank pads to the same row count (equal-size gather)
    gathered = np.ascontiguousarray(np.stack(parts))
    out = np.empty_like(whole)
    N.check(N.lib().vpt_probe_assemble_rows(gpu_ctx._h, gathered.ctypes.data_as(C.c_void_p), sc.w, sc.h, parts[0].shape[0], world, rows,
                                            out.ctypes.data_as(C.c_void_p)))
    assert_same_bits(out, whole, "assembled %d-rank frame" % world)
    # the torch host's index table describes the same permutation
    from vpt_amd.tiles import gather_index, local_rows
    idx = gather_index(sc.h, world, rows)
    flat = gathered.reshape(world * parts[0].shape[0], sc.w, 4)
    assert_same_bits(flat[idx], whole, "gather_index permutation")
    sc.gvol.destroy()
