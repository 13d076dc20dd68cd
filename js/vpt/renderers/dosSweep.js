'use strict';
// The host arithmetic of the DOS renderer's sweep (src/js/renderers/DOSRenderer.js:103-167,240-259), in this repository's own form; mirrored
// line for line by vpt_amd/dos_sweep.py.  The kernels are compared bit for bit, so what is kept from the reference is the ARITHMETIC — which
// doubles are rounded to float32 where, and the order of every sum — not its text:
//   * occlusion taps: n points of the unit disk (radius sqrt(u1), angle 2 pi u2, in draw order), re-centred on their centroid; the centroid is
//     the running sum of p / n in tap order, and a tap is rounded to float32 BEFORE the centroid is taken off it (the reference's Float32Array);
//   * the sweep's depth range: -z of the unit cube's corners in view space (float32 matrices, double points);
//   * a slice: the clip-space image of (1, 1, -depth) scaled by the occlusion cone's radius at one slice distance.
const { vec3, mat4 } = require('../math.js');

const TWO_PI = 2 * Math.PI;

function occlusionTaps(rng, count) {
    const points = [];
    for (let k = 0; k < count; k++) {
        const radius = Math.sqrt(rng());
        const angle = rng() * 2 * Math.PI;                  // (u * 2) * pi: the reference's order of the two products
        points.push([radius * Math.cos(angle), radius * Math.sin(angle)]);
    }
    const centroid = points.reduce((c, p) => [c[0] + p[0] / count, c[1] + p[1] / count], [0, 0]);
    const taps = new Float32Array(2 * count);
    points.forEach((p, k) => {
        taps[2 * k] = Math.fround(p[0]) - centroid[0];
        taps[2 * k + 1] = Math.fround(p[1]) - centroid[1];
    });
    return taps;
}

// [nearest, farthest] view-space depth of the volume: the unit cube goes through centre (-1/2), model and view matrix, each product a float32
// matrix as gl-matrix leaves it; the corners are transformed as double points
function viewDepthRange(modelMatrix, viewMatrix) {
    const centre = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]);
    const toView = [centre, modelMatrix, viewMatrix].reduce((m, factor) => mat4.multiply(m, factor, m), mat4.create());
    let nearest = Infinity, farthest = -Infinity;
    for (let corner = 0; corner < 8; corner++) {
        const p = [(corner >> 2) & 1, (corner >> 1) & 1, corner & 1];
        const depth = -vec3.transformMat4(p, p, toView)[2];
        nearest = Math.min(nearest, depth);
        farthest = Math.max(farthest, depth);
    }
    return [nearest, farthest];
}

// the (uOcclusionScale.x, uOcclusionScale.y, uDepth) triples of up to `count` slices from sweep.depth on; advances sweep.depth
function sliceTriples(sweep, count, sliceDistance, apertureDegrees, projectionMatrix) {
    const coneRadius = sliceDistance * Math.tan(apertureDegrees * Math.PI / 180);
    const triples = [];
    for (let k = 0; k < count && !(sweep.depth > sweep.farthest); k++) {
        const clip = vec3.transformMat4([0, 0, 0], [1, 1, -sweep.depth], projectionMatrix);
        triples.push(clip[0] * coneRadius, clip[1] * coneRadius, clip[2]);
        sweep.depth += sliceDistance;
    }
    return new Float32Array(triples);
}

module.exports = { occlusionTaps, viewDepthRange, sliceTriples, TWO_PI };
