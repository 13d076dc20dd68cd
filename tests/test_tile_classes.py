"""CPU: the tile classification behind k_mcm_miss (vpt_classify_tiles, host code of libvpt_hip.so — no GPU is touched).

A tile may be called MISS only if NO camera ray of its pixels (unprojectRand with blur == 0: from the pixel's near-plane point to a
far-plane point jittered by up to one pixel, mixins/unprojectRand.glsl:3-24) meets the unit cube.  Checked here by brute force in
float64: every pixel of every MISS tile, the un-jittered ray and the four extreme jitters, slab-tested against the cube
(mixins/intersectCube.glsl:3-11) — and that the classification is not uselessly timid (nearly every tile whose rays all miss by a
margin IS called MISS)."""
import ctypes as C

import numpy as np
import pytest

from vpt_amd import _native as N
from vpt_amd.scene import default_camera, Transform, Node, mvp_inverse_matrix

from conftest import orbit_camera


def classify(w, h, m, rank=0, world=1, rows=8):
    L = N.lib()
    tx, ty = C.c_int(0), C.c_int(0)
    m = np.ascontiguousarray(m, dtype=np.float32)
    N.check(L.vpt_classify_tiles(w, h, rank, world, rows, m.ctypes.data_as(C.c_void_p), None, 0, C.byref(tx), C.byref(ty)))
    cls = np.zeros(tx.value * ty.value, dtype=np.uint8)
    N.check(L.vpt_classify_tiles(w, h, rank, world, rows, m.ctypes.data_as(C.c_void_p), cls.ctypes.data_as(C.c_void_p), cls.size,
                                 C.byref(tx), C.byref(ty)))
    return cls.reshape(ty.value, tx.value)


def local_to_global_rows(h, rank, world, rows):
    if world == 1:
        return np.arange(h)
    nblocks = (h + rows - 1) // rows
    local_h = ((nblocks + world - 1) // world) * rows
    l = np.arange(local_h)
    lb = l // rows
    return (lb * world + rank) * rows + (l - lb * rows)          # >= h: padding


def rays_hit(w, h, m, cols, grows, jitter):
    """bool [len(grows)][len(cols)]: does the ray of pixel (col, global row) with NDC jitter (jx, jy) on the far plane meet [0,1]^3?"""
    M = np.asarray(m, dtype=np.float64).reshape(4, 4).T          # column-major -> M[row][col]
    px = (2.0 * cols + 1.0) / w - 1.0
    py = (2.0 * grows + 1.0) / h - 1.0
    X, Y = np.meshgrid(px, py)

    def unproject(x, y, z):
        v = np.stack([x, y, np.full_like(x, z), np.ones_like(x)], axis=-1) @ M.T
        return v[..., :3] / v[..., 3:4]
    f = unproject(X, Y, -1.0)
    t = unproject(X + jitter[0], Y + jitter[1], 1.0)
    d = t - f
    with np.errstate(divide='ignore', invalid='ignore'):
        t0 = (0.0 - f) / d
        t1 = (1.0 - f) / d
    tn = np.nanmax(np.minimum(t0, t1), axis=-1)
    tf = np.nanmin(np.maximum(t0, t1), axis=-1)
    return tf >= np.maximum(tn, 0.0)


CAMERAS = [
    ("default 16:9", lambda a: default_camera(a)),
    ("orbit", lambda a: orbit_camera(a)),
    ("orbit close", lambda a: orbit_camera(a, 2.2, 0.4, 1.2)),
    ("orbit far, off axis", lambda a: orbit_camera(a, -1.1, 0.9, 3.5)),
]


@pytest.mark.parametrize("name,cam", CAMERAS)
@pytest.mark.parametrize("size,shard", [((1920, 1080), (0, 1, 8)), ((1920, 1080), (3, 8, 8)), ((333, 207), (0, 1, 8)), ((333, 207), (1, 3, 5))])
def test_miss_tiles_hold_no_ray_that_meets_the_cube(name, cam, size, shard):
    w, h = size
    rank, world, rows = shard
    m = mvp_inverse_matrix(cam(w / h), Transform(Node()))
    cls = classify(w, h, m, rank, world, rows)
    ty, tx = cls.shape
    grows = local_to_global_rows(h, rank, world, rows)
    assert ty == (len(grows) + 15) // 16 and tx == (w + 15) // 16
    valid = grows < h
    cols = np.arange(w, dtype=np.float64)
    hit_any = np.zeros((len(grows), w), dtype=bool)
    jit = [(0.0, 0.0)] + [(sx / w, sy / h) for sx in (-1.0, 1.0) for sy in (-1.0, 1.0)]
    for j in jit:
        hit_any[valid] |= rays_hit(w, h, m, cols, grows[valid].astype(np.float64), j)
    # per tile: any pixel whose ray can meet the cube
    pad_r, pad_c = ty * 16 - len(grows), tx * 16 - w
    tiles_hit = np.pad(hit_any, ((0, pad_r), (0, pad_c))).reshape(ty, 16, tx, 16).any(axis=(1, 3))
    wrong = (cls == 1) & tiles_hit
    assert not wrong.any(), "%s: %d MISS tiles hold a ray that meets the cube" % (name, int(wrong.sum()))
    # usefulness: tiles without any hitting ray that are NOT called MISS are only a rim around the cube's silhouette
    timid = (cls == 0) & ~tiles_hit
    assert timid.sum() <= 0.12 * cls.size + 8, (name, int(timid.sum()), cls.size)
    if name == "default 16:9" and size == (1920, 1080):
        assert (cls == 1).mean() > 0.70                                  # the benchmark camera: ~3/4 of the image never meets the volume


def test_doubtful_matrices_classify_nothing_as_miss():
    w, h = 320, 200
    inside = mvp_inverse_matrix(orbit_camera(w / h, 0.3, 0.2, 0.35), Transform(Node()))        # camera INSIDE the cube
    assert classify(w, h, inside).sum() == 0
    assert classify(w, h, np.zeros(16, np.float32)).sum() == 0                                  # singular
    bad = np.array(mvp_inverse_matrix(default_camera(w / h), Transform(Node())), np.float32)
    bad[5] = np.nan
    assert classify(w, h, bad).sum() == 0
    # the volume behind the camera: no corner in front of the eye plane -> nothing is assumed
    cam = default_camera(w / h)
    cam.transform.localTranslation = [0, 0, -2]
    assert classify(w, h, mvp_inverse_matrix(cam, Transform(Node()))).sum() == 0


def test_volume_off_screen_is_all_miss():
    w, h = 320, 200
    cam = default_camera(w / h)
    cam.transform.localTranslation = [40, 0, 2]                          # looking down -z, far to the side of the cube
    cls = classify(w, h, mvp_inverse_matrix(cam, Transform(Node())))
    assert cls.all()


def test_a_distant_camera_classifies_nothing_as_miss():
    """The margin of the classification is absolute (1e-3 of the cube) while the kernels' fp32 error of from + t * direction grows with
    |from|: with the near plane thousands of units away (a distant, narrow camera standing in for an orthographic one) no tile may be
    called MISS; a camera at a moderate distance still is classified"""
    w, h = 320, 200
    for dist, expect_miss in ((30.0, True), (5000.0, False)):
        cam = default_camera(w / h)
        cam.transform.localTranslation = [0.3, 0.2, dist]
        pc = cam.components[0]
        pc.fovy = 0.02; pc.near = 0.5 * dist; pc.far = 2.0 * dist
        cls = classify(w, h, mvp_inverse_matrix(cam, Transform(Node())))
        assert bool(cls.any()) == expect_miss, (dist, int(cls.sum()))
