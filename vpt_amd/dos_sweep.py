"""The host arithmetic of the DOS renderer's sweep (src/js/renderers/DOSRenderer.js:103-167,240-259) — the Python mirror of
js/vpt/renderers/dosSweep.js, line for line.  The kernels are compared bit for bit, so what is kept from the reference is the arithmetic
(which doubles are rounded to float32 where, the order of every sum), not its text."""
import math

import numpy as np

from .scene import mat4, vec3


def occlusion_taps(rng, count):
    """count points of the unit disk (radius sqrt(u1), angle 2 pi u2, in draw order), re-centred on their centroid; float32 [2 * count]"""
    points = []
    for _ in range(count):
        radius = math.sqrt(rng())
        angle = rng() * 2 * math.pi                       # (u * 2) * pi: the reference's order of the two products
        points.append((radius * math.cos(angle), radius * math.sin(angle)))
    cx = cy = 0.0
    for x, y in points:                                   # the running sum of p / n in tap order
        cx += x / count
        cy += y / count
    taps = np.zeros(2 * count, dtype=np.float32)
    for k, (x, y) in enumerate(points):                   # a tap is rounded to float32 BEFORE the centroid is taken off it
        taps[2 * k] = float(np.float32(x)) - cx
        taps[2 * k + 1] = float(np.float32(y)) - cy
    return taps


def view_depth_range(model_matrix, view_matrix):
    """[nearest, farthest] view-space depth of the unit cube through centre (-1/2), model and view matrix (float32 products, double points)"""
    to_view = mat4.create()
    for factor in (mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]), model_matrix, view_matrix):
        mat4.multiply(to_view, factor, to_view)
    depths = []
    for corner in range(8):
        p = [(corner >> 2) & 1, (corner >> 1) & 1, corner & 1]
        depths.append(-vec3.transformMat4(p, p, to_view)[2])
    return [min(depths), max(depths)]


def slice_triples(sweep, count, slice_distance, aperture_degrees, projection_matrix):
    """the (uOcclusionScale.x, uOcclusionScale.y, uDepth) triples of up to `count` slices from sweep['depth'] on; advances sweep['depth']"""
    cone_radius = slice_distance * math.tan(aperture_degrees * math.pi / 180)
    triples = []
    for _ in range(count):
        if sweep['depth'] > sweep['farthest']:
            break
        clip = vec3.transformMat4([0, 0, 0], [1, 1, -sweep['depth']], projection_matrix)
        triples.append([clip[0] * cone_radius, clip[1] * cone_radius, clip[2]])
        sweep['depth'] += slice_distance
    return np.array(triples, dtype=np.float32).reshape(-1, 3)
