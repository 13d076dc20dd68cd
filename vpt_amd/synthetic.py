"""Deterministic synthetic inputs (no files): volumes, transfer functions, per-frame seeds."""
import numpy as np

VOLUME_SEED = 0x5EED0001


def sphere_volume(n, noise=0.0, seed=VOLUME_SEED, dims=None, z_range=None):
    """uint8 [z][y][x]: v = 255 * clamp(1 - |p - 1/2| / 0.45) (+ noise * smooth lattice noise), p at voxel centres.
    z_range = (z0, z1) returns only that slab of slices (large volumes are generated slab by slab)."""
    nz, ny, nx = dims if dims is not None else (n, n, n)
    z0, z1 = z_range if z_range is not None else (0, nz)
    out = np.empty((z1 - z0, ny, nx), dtype=np.uint8)
    x = ((np.arange(nx, dtype=np.float32) + 0.5) / nx - 0.5) ** 2
    y = ((np.arange(ny, dtype=np.float32) + 0.5) / ny - 0.5) ** 2
    lat = None
    if noise > 0.0:
        rng = np.random.Generator(np.random.PCG64(seed))
        lat = [rng.random((c, c, c), dtype=np.float32) for c in (9, 33)]
    xy = y[:, None] + x[None, :]
    for z in range(z0, z1):
        zz = ((z + 0.5) / nz - 0.5) ** 2
        v = 255.0 * np.clip(1.0 - np.sqrt(xy + np.float32(zz)) / 0.45, 0.0, 1.0)
        if lat is not None:
            v = v + noise * (_lattice_slice(lat[0], z, nz, ny, nx) * 0.65 + _lattice_slice(lat[1], z, nz, ny, nx) * 0.35 - 0.5) * (v > 0)
        out[z - z0] = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return out


def _lattice_slice(lat, z, nz, ny, nx):
    """trilinear sample of a small random lattice on slice z"""
    c = lat.shape[0] - 1
    fz = (z + 0.5) / nz * c
    z0 = min(int(fz), c - 1); tz = np.float32(fz - z0)
    sl = lat[z0] * (1 - tz) + lat[z0 + 1] * tz
    fy = (np.arange(ny, dtype=np.float32) + 0.5) / ny * c
    y0 = np.minimum(fy.astype(np.int64), c - 1); ty = (fy - y0)[:, None]
    rows = sl[y0] * (1 - ty) + sl[y0 + 1] * ty
    fx = (np.arange(nx, dtype=np.float32) + 0.5) / nx * c
    x0 = np.minimum(fx.astype(np.int64), c - 1); tx = (fx - x0)[None, :]
    return rows[:, x0] * (1 - tx) + rows[:, x0 + 1] * tx


def ramp_tf(width=256, rgb=(255, 255, 255)):
    """width x 1 RGBA8 ramp with alpha = v"""
    t = np.zeros((1, width, 4), dtype=np.uint8)
    a = np.rint(np.linspace(0, 255, width)).astype(np.uint8)
    t[0, :, 0] = (a.astype(np.uint16) * rgb[0] // 255).astype(np.uint8)
    t[0, :, 1] = (a.astype(np.uint16) * rgb[1] // 255).astype(np.uint8)
    t[0, :, 2] = (a.astype(np.uint16) * rgb[2] // 255).astype(np.uint8)
    t[0, :, 3] = a
    return t


def colour_tf(width=256, height=1, seed=7):
    """a smooth multi-colour RGBA8 transfer function (exercises sRGB decode and bilinear lookup)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    knots = rng.integers(0, 256, size=(height, 9, 4)).astype(np.float32)
    knots[:, 0, 3] = 0
    xs = np.linspace(0, 8, width)
    i0 = np.minimum(xs.astype(np.int64), 7); t = (xs - i0)[None, :, None]
    tf = knots[:, i0] * (1 - t) + knots[:, i0 + 1] * t
    return np.clip(np.rint(tf), 0, 255).astype(np.uint8)


class GoldenRatioRng:
    """seed_k = fract(k * 0.61803398875), k = 1, 2, ... — the per-frame 'Math.random()' of the fixed-seed runs"""

    def __init__(self, start=1):
        self.k = start

    def __call__(self):
        v = (self.k * 0.61803398875) % 1.0
        self.k += 1
        return v
