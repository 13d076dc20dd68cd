"""GPU: randomised differential test — random volumes (1..48 voxels per axis, non-cubic), image sizes (1..200 pixels, odd),
cameras (outside, inside, grazing, looking away), transfer functions (1..256 wide, 1..3 rows), environment maps,
filters and renderer parameters (zero extinction, zero bounces, one step, strong anisotropy ...), several frames
each — every buffer of every pass must equal the CPU oracle bit for bit.  Seeds are fixed: the cases never change."""
import math
import os

import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.scene import Node, Transform, PerspectiveCamera, quat, mvp_inverse_matrix, iso_light_direction
from vpt_amd.synthetic import GoldenRatioRng

pytestmark = pytest.mark.gpu

MCM_BUFFERS = [N.BUFFER_MCM_POSITION, N.BUFFER_MCM_DIRECTION, N.BUFFER_MCM_TRANSMITTANCE, N.BUFFER_MCM_RADIANCE]
KINDS = ["mip", "eam", "mcs", "mcm", "iso", "depth", "lao"]      # dos: tests/test_gpu_dos.py (slice sweep, own driver)


def same_bits(got, want, what):
    g = np.ascontiguousarray(got).view(np.uint8).reshape(-1); w = np.ascontiguousarray(want).view(np.uint8).reshape(-1)
    assert g.shape == w.shape, "%s: shapes %r vs %r" % (what, np.shape(got), np.shape(want))
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, "%s: %d of %d bytes differ, first at byte %d" % (what, bad.size, g.size, bad[0])


def random_camera(rng, aspect):
    """a camera somewhere around (or inside) the unit cube's model space [-0.5, 0.5]^3, random orientation and lens"""
    node = Node()
    mode = rng.integers(0, 4)
    if mode == 0:                                    # orbiting, looking roughly at the centre
        yaw, pitch, dist = rng.uniform(-math.pi, math.pi), rng.uniform(-1.2, 1.2), rng.uniform(0.9, 3.0)
    elif mode == 1:                                  # inside the volume
        yaw, pitch, dist = rng.uniform(-math.pi, math.pi), rng.uniform(-1.2, 1.2), rng.uniform(0.0, 0.45)
    elif mode == 2:                                  # close to a face: grazing rays
        yaw, pitch, dist = rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.5, 0.6)
    else:                                            # far away, narrow part of the image covered
        yaw, pitch, dist = rng.uniform(-math.pi, math.pi), rng.uniform(-0.5, 0.5), rng.uniform(4.0, 9.0)
    qy = quat.setAxisAngle(quat.create(), [0, 1, 0], yaw)
    qx = quat.setAxisAngle(quat.create(), [1, 0, 0], pitch)
    q = quat.multiply(quat.create(), qy, qx)
    if rng.uniform() < 0.25:                         # looking somewhere else entirely
        q = quat.multiply(quat.create(), q, quat.setAxisAngle(quat.create(), [0, 1, 0], rng.uniform(0.5, 2.5)))
    node.transform.localRotation = q
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    off = rng.uniform(-0.2, 0.2, size=3) if mode != 2 else np.zeros(3)
    node.transform.localTranslation = [dist * sy * cp + off[0], -dist * sp + off[1], dist * cy * cp + off[2]]
    cam = PerspectiveCamera(node, {'fovy': float(rng.uniform(0.3, 1.6)), 'aspect': aspect,
                                   'near': float(rng.choice([0.01, 0.1, 0.5])), 'far': float(rng.choice([10.0, 100.0]))})
    node.components.append(cam)
    return node


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(1, 49, size=3))                      # (nz, ny, nx)
    if seed % 7 == 0:
        dims = (1, 1, 1)
    style = rng.integers(0, 3)
    if style == 0:
        vol = rng.integers(0, 256, size=dims, dtype=np.uint8)
    elif style == 1:
        z, y, x = np.meshgrid(*[np.linspace(-1, 1, n) for n in dims], indexing='ij')
        vol = np.clip(255 * (1.0 - np.sqrt(x * x + y * y + z * z) / 1.1) + rng.normal(0, 12, size=dims), 0, 255).astype(np.uint8)
    else:
        vol = np.full(dims, int(rng.integers(0, 256)), dtype=np.uint8)
    if seed % 3 == 1:                                  # a two-channel (RG8) volume: the transfer function is looked up in 2-D
        second = rng.integers(0, 256, size=dims, dtype=np.uint8) if rng.uniform() < 0.7 else (255 - vol)
        vol = np.ascontiguousarray(np.stack([vol, second], axis=-1))
    elif seed % 5 == 2:                                # FLOAT texels (R32F, or an R16F file widened): the value itself, also outside [0, 1]
        frng = np.random.default_rng(77 + seed)        # (own generator: the other cases keep their draws)
        vol = (vol.astype(np.float32) / np.float32(255.0) * np.float32(frng.uniform(0.5, 1.6)) + np.float32(frng.uniform(-0.3, 0.2))).astype(np.float32)
        if frng.uniform() < 0.5:
            vol = vol.astype(np.float16).astype(np.float32)
        if seed % 10 == 7:                             # two float channels (RG32F): both filtered, the transfer function looked up in 2-D
            second = (frng.uniform(-0.2, 1.3, size=vol.shape)).astype(np.float32)
            vol = np.ascontiguousarray(np.stack([vol, second], axis=-1))
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 140))
    tf_w, tf_h = int(rng.choice([1, 2, 3, 7, 64, 256])), int(rng.choice([1, 1, 3, 16]))
    tf = rng.integers(0, 256, size=(tf_h, tf_w, 4), dtype=np.uint8)
    if rng.uniform() < 0.3:
        tf = None
    env = rng.integers(0, 256, size=(int(rng.integers(1, 6)), int(rng.integers(1, 9)), 4), dtype=np.uint8) if rng.uniform() < 0.4 else None
    filt = "nearest" if rng.uniform() < 0.3 else "linear"
    model = Transform(Node())
    if rng.uniform() < 0.5:
        model.localScale = [float(v) for v in rng.uniform(0.5, 1.8, size=3)]
        model.localTranslation = [float(v) for v in rng.uniform(-0.2, 0.2, size=3)]
        model.localRotation = quat.setAxisAngle(quat.create(), [0.3, 0.8, 0.52], float(rng.uniform(-1, 1)))
    return rng, vol, (w, h), tf, env, filt, model


# VPT_FUZZ_SEEDS=a:b widens the sweep (e.g. 40:400); the default 40 seeds keep the suite short
_SEEDS = range(*[int(v) for v in os.environ.get("VPT_FUZZ_SEEDS", "0:40").split(":")])
# VPT_FUZZ_SPLIT=K: every case with VPT_OPTION_SPLIT_STREAMS = K; 1: none; default (-1): every third seed with three streams
_SPLIT = int(os.environ.get("VPT_FUZZ_SPLIT", "-1"))
# VPT_FUZZ_LAZY=1: every case compares its buffers after the LAST pass only (0: after every pass); default (-1): every third seed.
# Between two reads the MCM tile classes leave the position / transmittance arrays of the cube-missing tiles behind and the ray marchers
# launch the cube-crossing tiles only: a lazy case checks that what is read in the end is right all the same.
_LAZY = int(os.environ.get("VPT_FUZZ_LAZY", "-1"))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("seed", _SEEDS)
def test_random_scene(gpu_ctx, oracle, kind, seed):
    rng, vol, (w, h), tf, env, filt, model = random_case(seed * 6 + KINDS.index(kind) if kind != "lao" else 5000 + seed)
    camera = random_camera(rng, w / h)
    m = mvp_inverse_matrix(camera, model)
    osc = oracle.OracleScene(vol, filt, tf=tf, env=env)
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, filt)
    fused = bool(rng.integers(0, 2))
    start = int(rng.integers(1, 50))
    opts = {'resolution': (w, h), 'transform': model, 'rng': GoldenRatioRng(start), 'fused': fused}
    if rng.uniform() < 0.3:                           # one rank of a row-sharded run: its rows must be the oracle's rows
        world = int(rng.integers(2, 6))
        opts['shard'] = (int(rng.integers(0, world)), world, int(rng.choice([1, 3, 8, 16])))
    r = vpt_amd.RendererFactory(kind)(gpu_ctx, gvol, camera, env, opts)
    if _SPLIT >= 2 or (_SPLIT < 0 and seed % 3 == 1):          # a third of the default cases run their passes on three streams
        r.set_option(N.OPTION_SPLIT_STREAMS, max(_SPLIT, 3) if _SPLIT >= 2 else 3)
    rows = r.global_rows()
    valid = rows >= 0

    def same_rows(got, want, msg):
        same_bits(np.ascontiguousarray(got).reshape(len(rows), -1)[valid], np.ascontiguousarray(want).reshape(h, -1)[rows[valid]], msg)

    if tf is not None:
        r.setTransferFunction(tf)
    if kind == "mip":
        r.steps = int(rng.choice([1, 2, 3, 17, 64, 100]))
    elif kind == "eam":
        r.slices = int(rng.choice([1, 5, 33, 64])); r.extinction = float(rng.choice([0.0, 1.0, 40.0, 300.0])); r.random = bool(rng.integers(0, 2))
    elif kind == "mcs":
        r.extinction = float(rng.choice([0.0, 0.5, 5.0, 60.0]))
    elif kind == "mcm":
        r.extinction = float(rng.choice([0.0, 1.0, 7.0, 80.0])); r.anisotropy = float(rng.choice([0.0, 0.0, 0.9, -0.7, 1e-6]))
        r.bounces = int(rng.choice([0, 1, 8])); r.steps = int(rng.choice([1, 3, 8]))
    elif kind == "iso":
        r.steps = int(rng.choice([1, 2, 7, 50])); r.isovalue = float(rng.choice([0.0, 0.2, 0.5, 1.0])); r.light = [float(v) for v in rng.uniform(-3, 3, size=3)]
    elif kind == "depth":
        r.slices = int(rng.choice([1, 9, 64])); r.extinction = float(rng.choice([0.0, 10.0, 100.0])); r.threshold = float(rng.choice([0.0, 0.1, 0.9])); r.random = bool(rng.integers(0, 2))
    elif kind == "lao":
        r.slices = int(rng.choice([1, 7, 24])); r.extinction = float(rng.choice([0.0, 30.0, 100.0, 400.0]))
        r.localAmbientOcclusion = bool(rng.integers(0, 2)); r.softShadows = bool(rng.integers(0, 2))
        r.LAOWeight = float(rng.uniform(0, 1)); r.shadowsWeight = float(rng.uniform(0, 1))
        r.numLAOSamples = int(rng.integers(1, 4)); r.numShadowSamples = int(rng.integers(1, 12))
        r.LAOStepSize = float(rng.choice([0.01, 0.05, 0.3, 1.5])); r.lightRadious = float(rng.choice([0.0, 0.19, 0.7]))
        r.lightCoeficient = float(rng.choice([0.5, 1.0, 3.0])); r.lightPosition = [float(v) for v in rng.uniform(-12, 12, size=3)]
    o = oracle.OracleRenderer(kind, osc, w, h)
    if kind == "lao":
        o.lao = oracle.lao_params(local_ambient_occlusion=int(r.localAmbientOcclusion), lao_weight=r.LAOWeight, num_lao_samples=r.numLAOSamples,
                                  lao_step_size=r.LAOStepSize, soft_shadows=int(r.softShadows), shadows_weight=r.shadowsWeight,
                                  num_shadow_samples=r.numShadowSamples, light_radius=r.lightRadious, light_coefficient=r.lightCoeficient,
                                  light_position=r.lightPosition)
    what = "%s seed %d (%dx%d image, volume %s, %s, fused=%s, shard=%s)" % (kind, seed, w, h, vol.shape, filt, fused, opts.get('shard'))

    def frame_of(u):
        fr = oracle.make_frame(w, h, np.array(list(u.mvp_inverse), np.float32))
        fr.seed = u.rand_seed; fr.offset = u.offset; fr.step = u.step_size
        fr.extinction = u.extinction; fr.anisotropy = u.anisotropy; fr.max_bounces = u.max_bounces; fr.steps = u.steps
        for i in range(3):
            fr.light_dir[i] = u.light_direction[i]
        fr.mix = u.mix; fr.blur = u.blur
        fr.isovalue = u.isovalue; fr.gradient_step = u.gradient_step; fr.threshold = u.threshold
        return fr

    r.reset()
    if kind == "mcm":
        o.reset(oracle.make_frame(w, h, m, seed=np.float32(GoldenRatioRng(start)())))      # MCMRenderer.js:93: the reset's own draw
        for b, s in zip(MCM_BUFFERS, o.state):
            same_rows(r.read(b), s, what + " reset buffer %d" % b)
    else:
        o.reset(oracle.make_frame(w, h, m))
    lazy = _LAZY == 1 or (_LAZY < 0 and seed % 3 == 2)
    npasses = 5 if lazy else 3
    for k in range(npasses):
        r.render()
        o.render(frame_of(r._u))
        if lazy and k + 1 < npasses:
            continue
        if kind == "mcm":
            for b, s in zip(MCM_BUFFERS, o.state):
                same_rows(r.read(b), s, what + " state %d pass %d" % (b, k))
        else:
            if not fused:
                same_rows(r.read(N.BUFFER_FRAME), o.frame, what + " frame %d" % k)
            same_rows(r.read(N.BUFFER_ACCUM), o.acc, what + " accumulation %d" % k)
        same_rows(r.getTexture().view(np.uint16), o.out, what + " render %d" % k)
    if 'shard' not in opts:
        assert r.sample_count() == o.samples, what
    r.destroy(); gvol.destroy()


@pytest.mark.parametrize("w,h", [(5000, 2), (2, 5000), (4099, 33), (1, 1), (17, 4097)])
@pytest.mark.parametrize("kind", ["mip", "mcm"])
def test_extreme_image_shapes(gpu_ctx, oracle, kind, w, h):
    """very wide / very tall / single-pixel images: the tile -> workgroup map at its edges (tiles_x not a multiple of 8,
    hundreds of tile rows, one-wave and four-wave workgroups)"""
    rng = np.random.default_rng(w * 31 + h)
    vol = rng.integers(0, 256, size=(9, 11, 13), dtype=np.uint8)
    camera = random_camera(np.random.default_rng(5), w / h)
    model = Transform(Node())
    m = mvp_inverse_matrix(camera, model)
    osc = oracle.OracleScene(vol, "linear")
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, "linear")
    r = vpt_amd.RendererFactory(kind)(gpu_ctx, gvol, camera, None, {'resolution': (w, h), 'transform': model, 'rng': GoldenRatioRng()})
    o = oracle.OracleRenderer(kind, osc, w, h)
    r.reset()
    if kind == "mcm":
        r.extinction = 5
        o.reset(oracle.make_frame(w, h, m, seed=np.float32(GoldenRatioRng()())))
    else:
        o.reset(oracle.make_frame(w, h, m))
    rg = GoldenRatioRng(2 if kind == "mcm" else 1)
    for k in range(2):
        r.render()
        if kind == "mcm":
            o.render(oracle.make_frame(w, h, m, seed=np.float32(rg()), extinction=5, nthreads=4))
        else:
            o.render(oracle.make_frame(w, h, m, offset=np.float32(rg()), steps=64, nthreads=4))
    same_bits(r.getTexture().view(np.uint16), o.out, "%s %dx%d" % (kind, w, h))
    assert r.sample_count() == o.samples
    r.destroy(); gvol.destroy()


@pytest.mark.parametrize("dims", [(2, 3, 4096), (4096, 2, 3), (3, 4096, 1), (1, 1, 4096)])
def test_extreme_volume_shapes(gpu_ctx, oracle, dims):
    """a 4096-voxel axis (the largest the C ABI accepts) next to 1..3-voxel axes: offset tables, brick padding, LDS size"""
    rng = np.random.default_rng(sum(dims))
    vol = rng.integers(0, 256, size=dims, dtype=np.uint8)                       # (nz, ny, nx)
    w, h = 90, 70
    camera = random_camera(np.random.default_rng(11), w / h)
    model = Transform(Node())
    m = mvp_inverse_matrix(camera, model)
    for filt in ("linear", "nearest"):
        osc = oracle.OracleScene(vol, filt)
        gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, filt)
        r = vpt_amd.EAMRenderer(gpu_ctx, gvol, camera, None, {'resolution': (w, h), 'transform': model, 'rng': GoldenRatioRng()})
        r.slices = 200
        o = oracle.OracleRenderer('eam', osc, w, h)
        r.reset(); o.reset(oracle.make_frame(w, h, m))
        r.render()
        o.render(oracle.make_frame(w, h, m, offset=np.float32(GoldenRatioRng()()), steps=200, extinction=100, mix=1.0))
        same_bits(r.getTexture().view(np.uint16), o.out, "eam volume %s %s" % (dims, filt))
        assert r.sample_count() == o.samples
        r.destroy(); gvol.destroy()


@pytest.mark.parametrize("tf_w", [1024, 2048])
@pytest.mark.parametrize("kind", ["eam", "mcm"])
def test_widest_transfer_functions(gpu_ctx, oracle, kind, tf_w):
    """transfer functions of 1024 and 2048 entries (the C ABI's limit): 32 / 64 KiB of (value, difference) pairs in LDS next to
    the offset tables — beyond the default 64 KiB dynamic-LDS limit for the widest one"""
    rng = np.random.default_rng(tf_w)
    vol = rng.integers(0, 256, size=(20, 24, 28), dtype=np.uint8)
    tf = rng.integers(0, 256, size=(1, tf_w, 4), dtype=np.uint8)
    w, h = 64, 48
    camera = random_camera(np.random.default_rng(3), w / h)
    model = Transform(Node())
    m = mvp_inverse_matrix(camera, model)
    osc = oracle.OracleScene(vol, "linear", tf=tf)
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, "linear")
    r = vpt_amd.RendererFactory(kind)(gpu_ctx, gvol, camera, None, {'resolution': (w, h), 'transform': model, 'rng': GoldenRatioRng()})
    r.setTransferFunction(tf)
    o = oracle.OracleRenderer(kind, osc, w, h)
    r.reset()
    if kind == "mcm":
        r.extinction = 6
        o.reset(oracle.make_frame(w, h, m, seed=np.float32(GoldenRatioRng()())))
        r.render()
        o.render(oracle.make_frame(w, h, m, seed=np.float32(GoldenRatioRng(2)()), extinction=6))
    else:
        o.reset(oracle.make_frame(w, h, m))
        r.render()
        o.render(oracle.make_frame(w, h, m, offset=np.float32(GoldenRatioRng()()), steps=64, extinction=100, mix=1.0))
    same_bits(r.getTexture().view(np.uint16), o.out, "%s tf %d" % (kind, tf_w))
    r.destroy(); gvol.destroy()


@pytest.mark.parametrize("kind", ["mcm", "mcs", "mip"])
def test_rg8_volume_with_frame_sequences_and_shards(gpu_ctx, oracle, kind):
    """the RG8 kernel variants through the other launch paths: fused passes, graph replay, a row shard"""
    rng = np.random.default_rng(77)
    vol = rng.integers(0, 256, size=(18, 20, 22, 2), dtype=np.uint8)
    tf = rng.integers(0, 256, size=(6, 16, 4), dtype=np.uint8)
    w, h = 88, 60
    camera = random_camera(np.random.default_rng(9), w / h)
    model = Transform(Node())
    gvol = vpt_amd.Volume.from_array(gpu_ctx, vol, "linear")

    def make(**opts):
        o = {'resolution': (w, h), 'transform': model, 'rng': GoldenRatioRng()}
        o.update(opts)
        r = vpt_amd.RendererFactory(kind)(gpu_ctx, gvol, camera, None, o)
        r.setTransferFunction(tf)
        if kind in ("mcm", "mcs"):
            r.extinction = 7
        r.reset()
        return r

    ref = make()
    for _ in range(6):
        ref.render()
    want = ref.getTexture()
    a = make(); a.play(6, fused=True)
    same_bits(a.getTexture().view(np.uint16), want.view(np.uint16), "%s RG8 fused passes" % kind)
    b = make(); b.render(); b.render(); b.play(4, use_graph=True)
    same_bits(b.getTexture().view(np.uint16), want.view(np.uint16), "%s RG8 graph replay" % kind)
    c = make(shard=(1, 3, 8))
    for _ in range(6):
        c.render()
    rows = c.global_rows(); valid = rows >= 0
    same_bits(c.getTexture()[valid].view(np.uint16), want[rows[valid]].view(np.uint16), "%s RG8 shard" % kind)
    # and against the oracle
    osc = oracle.OracleScene(vol, "linear", tf=tf)
    m = mvp_inverse_matrix(camera, model)
    o = oracle.OracleRenderer(kind, osc, w, h)
    rg = GoldenRatioRng()
    if kind == "mcm":
        o.reset(oracle.make_frame(w, h, m, seed=np.float32(rg())))
    else:
        o.reset(oracle.make_frame(w, h, m))
    r2 = make()
    for k in range(6):
        r2.render()
        u = r2._u
        fr = oracle.make_frame(w, h, np.array(list(u.mvp_inverse), np.float32))
        fr.seed = u.rand_seed; fr.offset = u.offset; fr.step = u.step_size; fr.extinction = u.extinction; fr.anisotropy = u.anisotropy
        fr.max_bounces = u.max_bounces; fr.steps = u.steps; fr.mix = u.mix; fr.blur = u.blur
        for i in range(3):
            fr.light_dir[i] = u.light_direction[i]
        o.render(fr)
    same_bits(want.view(np.uint16), o.out, "%s RG8 vs oracle" % kind)
    for r in (ref, a, b, c, r2):
        r.destroy()
    gvol.destroy()
