"""GPU: volumes that arrive through the readers (BVP blocks with partial x-y extents, file-backed loaders) land in HBM
exactly as a whole-array upload does (Volume.readModality's texSubImage3D placements, Volume.js:63-71)."""
import io
import json
import zipfile

import numpy as np
import pytest

import vpt_amd
from vpt_amd import _native as N
from vpt_amd.loaders import BlobLoader, FileLoader
from vpt_amd.readers import BVPReader, RAWReader
from vpt_amd.scene import Transform, Node, default_camera
from vpt_amd.synthetic import sphere_volume, GoldenRatioRng

pytestmark = pytest.mark.gpu


def make_bvp(vol, cuts):
    """vol [z][y][x] u8 -> stored-zip BVP bytes, cut into blocks at the given (x, y, z) split points"""
    d, h, w = vol.shape
    xs, ys, zs = ([0] + list(c) + [n] for c, n in zip(cuts, (w, h, d)))
    blocks, placements = [], []
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", compression=zipfile.ZIP_STORED) as z:
        for zi in range(len(zs) - 1):
            for yi in range(len(ys) - 1):
                for xi in range(len(xs) - 1):
                    x0, x1, y0, y1, z0, z1 = xs[xi], xs[xi + 1], ys[yi], ys[yi + 1], zs[zi], zs[zi + 1]
                    name = "blocks/%d_%d_%d.raw" % (xi, yi, zi)
                    z.writestr(name, np.ascontiguousarray(vol[z0:z1, y0:y1, x0:x1]).tobytes())
                    placements.append({"index": len(blocks), "position": {"x": x0, "y": y0, "z": z0}})
                    blocks.append({"url": name, "format": "raw", "dimensions": {"width": x1 - x0, "height": y1 - y0, "depth": z1 - z0}})
        manifest = {"meta": {"version": 1},
                    "modalities": [{"name": "default", "dimensions": {"width": w, "height": h, "depth": d},
                                    "transform": {"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]},
                                    "format": 6403, "internalFormat": 33321, "type": 5121, "placements": placements}],
                    "blocks": blocks}
        z.writestr("manifest.json", json.dumps(manifest))
    return bio.getvalue()


def render_mip(ctx, gvol, w=120, h=90, frames=2):
    r = vpt_amd.MIPRenderer(ctx, gvol, default_camera(w / h), None, {'resolution': (w, h), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
    r.steps = 50
    r.reset()
    for _ in range(frames):
        r.render()
    out = (r.read(N.BUFFER_ACCUM), r.getTexture())
    r.destroy()
    return out


@pytest.mark.parametrize("loader_kind", ["blob", "file"])
def test_bvp_volume_equals_whole_array_upload(gpu_ctx, tmp_path, loader_kind):
    vol = sphere_volume(0, noise=35.0, dims=(37, 45, 52))                       # odd sizes: blocks end mid-brick
    archive = make_bvp(vol, cuts=((20, 33), (17,), (5, 30)))                   # 3 x 2 x 3 blocks, partial x-y extents
    if loader_kind == "file":
        p = tmp_path / "v.bvp"; p.write_bytes(archive)
        loader = FileLoader(str(p))
    else:
        loader = BlobLoader(archive)
    seen = []
    v = vpt_amd.Volume(gpu_ctx, BVPReader(loader))
    v.addEventListener('progress', lambda e: seen.append(e.detail))
    assert v.getTexture() is None                                                # Volume.js:107-113
    v.load()
    v.setFilter('linear')
    assert v.ready and len(seen) == 18 and seen[-1] == 1
    whole = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')
    a, b = render_mip(gpu_ctx, v), render_mip(gpu_ctx, whole)
    assert (a[0] == b[0]).all() and (a[1].view(np.uint16) == b[1].view(np.uint16)).all()
    with pytest.raises(RuntimeError, match="Modality 'nope' does not exist"):   # Volume.js:40
        v.readModality('nope')
    v.destroy(); whole.destroy()


def test_raw_reader_from_file(gpu_ctx, tmp_path):
    vol = sphere_volume(0, noise=20.0, dims=(20, 24, 28))
    p = tmp_path / "v.raw"; p.write_bytes(vol.tobytes())
    v = vpt_amd.Volume(gpu_ctx, RAWReader(FileLoader(str(p)), {'width': 28, 'height': 24, 'depth': 20}))
    v.load(); v.setFilter('linear')
    whole = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')
    a, b = render_mip(gpu_ctx, v), render_mip(gpu_ctx, whole)
    assert (a[0] == b[0]).all()
    v.destroy(); whole.destroy()


def test_rg8_volume_through_bvp(gpu_ctx, oracle, tmp_path):
    """a two-channel BVP modality (format RG, internalFormat RG8): blocks of interleaved bytes land where texSubImage3D puts
    them, texture(uVolume, p).rg feeds a 2-D transfer function; every renderer equals the oracle on the same data"""
    rng = np.random.default_rng(21)
    vol = np.stack([sphere_volume(0, noise=35.0, dims=(22, 30, 26)), rng.integers(0, 256, size=(22, 30, 26), dtype=np.uint8)], axis=-1)
    d, h, w = vol.shape[:3]
    # two blocks along x with partial extents, each [bd][bh][bw][2] bytes
    blocks = [(0, 11), (11, w)]
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", compression=zipfile.ZIP_STORED) as z:
        for i, (x0, x1) in enumerate(blocks):
            z.writestr("b%d.raw" % i, np.ascontiguousarray(vol[:, :, x0:x1]).tobytes())
        manifest = {"meta": {"version": 1},
                    "modalities": [{"name": "default", "dimensions": {"width": w, "height": h, "depth": d},
                                    "transform": {"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]},
                                    "format": 33319, "internalFormat": 33323, "type": 5121,
                                    "placements": [{"index": i, "position": {"x": x0, "y": 0, "z": 0}} for i, (x0, x1) in enumerate(blocks)]}],
                    "blocks": [{"url": "b%d.raw" % i, "format": "raw", "dimensions": {"width": x1 - x0, "height": h, "depth": d}} for i, (x0, x1) in enumerate(blocks)]}
        z.writestr("manifest.json", json.dumps(manifest))
    v = vpt_amd.Volume(gpu_ctx, BVPReader(BlobLoader(bio.getvalue())))
    v.load(); v.setFilter('linear')
    whole = vpt_amd.Volume.from_array(gpu_ctx, vol, 'linear')
    tf = rng.integers(0, 256, size=(9, 32, 4), dtype=np.uint8)          # 32 x 9: both axes matter
    from conftest import default_matrix
    W, H = 96, 72
    m = default_matrix(W / H)
    osc = oracle.OracleScene(vol, 'linear', tf=tf)
    for kind in ('mip', 'eam', 'mcm'):
        outs = []
        for gv in (v, whole):
            r = vpt_amd.RendererFactory(kind)(gpu_ctx, gv, default_camera(W / H), None, {'resolution': (W, H), 'transform': Transform(Node()), 'rng': GoldenRatioRng()})
            r.setTransferFunction(tf)
            if kind == 'mcm':
                r.extinction = 9
            r.reset()
            for _ in range(2):
                r.render()
            outs.append(r.getTexture()); u = r._u
            r.destroy()
        assert (outs[0].view(np.uint16) == outs[1].view(np.uint16)).all(), kind
        o = oracle.OracleRenderer(kind, osc, W, H)
        rg = GoldenRatioRng()
        if kind == 'mcm':
            o.reset(oracle.make_frame(W, H, m, seed=np.float32(rg())))
            for _ in range(2):
                o.render(oracle.make_frame(W, H, m, seed=np.float32(rg()), extinction=9))
        else:
            o.reset(oracle.make_frame(W, H, m))
            for k in range(2):
                off = np.float32(rg())
                o.render(oracle.make_frame(W, H, m, offset=off, steps=64, extinction=100, mix=np.float32(1.0 / (k + 1))))
        assert (outs[0].view(np.uint16).reshape(-1) == o.out).all(), "%s vs oracle" % kind
    v.destroy(); whole.destroy()


def make_bvp_typed(arr, fmt, ifmt, gltype, cuts):
    """arr [z][y][x] or [z][y][x][c] of any dtype -> stored-zip BVP with the given GL format / internalFormat / type"""
    d, h, w = arr.shape[:3]
    xs, ys, zs = ([0] + list(c) + [n] for c, n in zip(cuts, (w, h, d)))
    blocks, placements = [], []
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, "w", compression=zipfile.ZIP_STORED) as z:
        for zi in range(len(zs) - 1):
            for yi in range(len(ys) - 1):
                for xi in range(len(xs) - 1):
                    x0, x1, y0, y1, z0, z1 = xs[xi], xs[xi + 1], ys[yi], ys[yi + 1], zs[zi], zs[zi + 1]
                    name = "blocks/%d_%d_%d.raw" % (xi, yi, zi)
                    z.writestr(name, np.ascontiguousarray(arr[z0:z1, y0:y1, x0:x1]).tobytes())
                    placements.append({"index": len(blocks), "position": {"x": x0, "y": y0, "z": z0}})
                    blocks.append({"url": name, "format": "raw", "dimensions": {"width": x1 - x0, "height": y1 - y0, "depth": z1 - z0}})
        manifest = {"meta": {"version": 1},
                    "modalities": [{"name": "default", "dimensions": {"width": w, "height": h, "depth": d},
                                    "transform": {"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]},
                                    "format": fmt, "internalFormat": ifmt, "type": gltype, "placements": placements}],
                    "blocks": blocks}
        z.writestr("manifest.json", json.dumps(manifest))
    return bio.getvalue()


@pytest.mark.parametrize("case", ["r32f", "r16f", "rgba8", "rgb8", "rg32f", "rgba16f"])
def test_float_and_multichannel_manifests(gpu_ctx, oracle, case):
    """Volume.js:58-60 allocates whatever internalFormat the manifest names, :84-105 maps the GL type: R32F / R16F volumes
    (FLOAT / HALF_FLOAT, filtered LINEAR), RGBA8 / RGB8 (the shaders read .rg) and two- / four-channel float volumes (RG32F;
    RGBA16F keeps its first two channels, widened) through a BVP with partial blocks must render exactly like the same texels
    uploaded as one array — and like the oracle"""
    from vpt_amd.readers import GL_RED, GL_R32F, GL_R16F, GL_FLOAT, GL_HALF_FLOAT, GL_RGBA, GL_RGBA8, GL_RGB, GL_RGB8, GL_UNSIGNED_BYTE
    rng = np.random.default_rng(11)
    base = sphere_volume(0, noise=35.0, dims=(19, 26, 23))
    if case in ("r32f", "r16f"):
        f = (base.astype(np.float32) / np.float32(255) * np.float32(1.4) - np.float32(0.2)).astype(np.float32)
        if case == "r16f":
            stored = f.astype(np.float16); texels = stored.astype(np.float32)
            archive = make_bvp_typed(stored, GL_RED, GL_R16F, GL_HALF_FLOAT, ((9,), (11, 20), (7,)))
        else:
            stored = texels = f
            archive = make_bvp_typed(stored, GL_RED, GL_R32F, GL_FLOAT, ((9,), (11, 20), (7,)))
    elif case in ("rg32f", "rgba16f"):
        nch = 2 if case == "rg32f" else 4
        f = (base.astype(np.float32) / np.float32(255) * np.float32(1.3) - np.float32(0.1)).astype(np.float32)
        stored = rng.uniform(-0.2, 1.2, size=base.shape + (nch,)).astype(np.float32)
        stored[..., 0] = f
        if case == "rgba16f":
            stored = stored.astype(np.float16)
            archive = make_bvp_typed(stored, GL_RGBA, 0x881A, GL_HALF_FLOAT, ((9,), (11, 20), (7,)))        # RGBA16F
        else:
            archive = make_bvp_typed(stored, 0x8227, 0x8230, GL_FLOAT, ((9,), (11, 20), (7,)))              # RG, RG32F
        texels = np.ascontiguousarray(stored[..., :2].astype(np.float32))
    else:
        nch = 4 if case == "rgba8" else 3
        stored = rng.integers(0, 256, size=base.shape + (nch,), dtype=np.uint8)
        stored[..., 0] = base
        texels = np.ascontiguousarray(stored[..., :2])
        archive = make_bvp_typed(stored, GL_RGBA if nch == 4 else GL_RGB, GL_RGBA8 if nch == 4 else GL_RGB8, GL_UNSIGNED_BYTE, ((9,), (11, 20), (7,)))
    v1 = vpt_amd.Volume(gpu_ctx, BVPReader(BlobLoader(archive))); v1.load(); v1.setFilter('linear')
    v2 = vpt_amd.Volume.from_array(gpu_ctx, texels, 'linear')
    a, ta = render_mip(gpu_ctx, v1); b, tb = render_mip(gpu_ctx, v2)
    assert (a == b).all() and (ta.view(np.uint16) == tb.view(np.uint16)).all() and a.max() > 0
    # against the oracle's linear-layout sampler (same seeds as render_mip)
    w, h = 120, 90
    from vpt_amd.scene import mvp_inverse_matrix
    m = mvp_inverse_matrix(default_camera(w / h), Transform(Node()))
    o = oracle.OracleRenderer('mip', oracle.OracleScene(texels, 'linear'), w, h)
    o.reset(oracle.make_frame(w, h, m))
    g = GoldenRatioRng()
    for _ in range(2):
        o.render(oracle.make_frame(w, h, m, steps=50, offset=np.float32(g())))
    assert (a.reshape(-1) == o.acc).all()
    v1.destroy(); v2.destroy()


def test_unsupported_gl_types_raise_the_reference_error(gpu_ctx):
    from vpt_amd.readers import GL_RED
    vol = np.zeros((4, 4, 4), np.uint16)
    for gltype in (5123, 5125, 5120, 5122):            # UNSIGNED_SHORT, UNSIGNED_INT, BYTE, SHORT: not filterable through sampler3D
        archive = make_bvp_typed(vol, GL_RED, 33322, gltype, ((), (), ()))
        v = vpt_amd.Volume(gpu_ctx, BVPReader(BlobLoader(archive)))
        with pytest.raises(RuntimeError, match="Unknown volume datatype"):
            v.load()
