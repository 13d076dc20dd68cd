// Generates tests/golden/mvp_inverse.json by running the reference's VENDORED gl-matrix 3.4.1
// (/root/reference/src/lib/gl-matrix-min.js, UMD, loads under node 12) through the exact call
// sequence of MIPRenderer.js:86-97 / Transform.js:26-29,64-70 / PerspectiveCamera.js:15-17.
// Run in the authoring container only:  node tests/golden/make_mvp_fixture.js
// The fixture holds INPUTS and OUTPUT BITS only; no reference source travels.
'use strict';
const fs = require('fs');
const path = require('path');
const glm = require('/root/reference/src/lib/gl-matrix-min.js');
const { mat4, quat, vec3 } = glm;

function bits(m) { return Array.from(new Uint32Array(new Float32Array(m).buffer)); }

// Transform.js:6-8,98-116: the setters store Float32Array clones (quat.clone / vec3.clone)
function stored(t) {
    return { rotation: quat.clone(t.rotation), translation: vec3.clone(t.translation), scale: vec3.clone(t.scale) };
}
// Transform.js:26-29
function localMatrix(t0) {
    const t = stored(t0);
    return mat4.fromRotationTranslationScale(mat4.create(), t.rotation, t.translation, t.scale);
}
// Transform.js:64-70
function inverseLocalMatrix(t0) {
    const t = stored(t0);
    const matrix = mat4.create();
    mat4.scale(matrix, matrix, vec3.inverse(vec3.create(), t.scale));
    mat4.multiply(matrix, matrix, mat4.fromQuat(mat4.create(), quat.invert(quat.create(), t.rotation)));
    mat4.translate(matrix, matrix, vec3.negate(vec3.create(), t.translation));
    return matrix;
}
// MIPRenderer.js:86-97
function mvpInverse(c) {
    const centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]);
    const modelMatrix = localMatrix(c.model);
    const viewMatrix = inverseLocalMatrix(c.camera);
    const projectionMatrix = mat4.perspective(mat4.create(), c.fovy, c.aspect, c.near, c.far);
    const matrix = mat4.create();
    mat4.multiply(matrix, centerMatrix, matrix);
    mat4.multiply(matrix, modelMatrix, matrix);
    mat4.multiply(matrix, viewMatrix, matrix);
    mat4.multiply(matrix, projectionMatrix, matrix);
    const forward = Array.from(matrix);
    mat4.invert(matrix, matrix);
    return { forward: forward, inverse: matrix };
}

// ISORenderer.js:152-166: the light direction (view space) taken into model space and normalised
function isoLight(c, light) {
    const centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]);
    const modelMatrix = localMatrix(c.model);
    const viewMatrix = inverseLocalMatrix(c.camera);
    const matrix = mat4.create();
    mat4.multiply(matrix, centerMatrix, matrix);
    mat4.multiply(matrix, modelMatrix, matrix);
    mat4.multiply(matrix, viewMatrix, matrix);
    mat4.invert(matrix, matrix);
    const l = vec3.transformMat4(vec3.create(), light, matrix);
    vec3.normalize(l, l);
    return l;
}

function qAxis(axis, rad) { return Array.from(quat.setAxisAngle(quat.create(), axis, rad)); }
const ident = { rotation: [0, 0, 0, 1], translation: [0, 0, 0], scale: [1, 1, 1] };
const cases = [
    { name: 'default_aspect1', fovy: 1, aspect: 1, near: 0.1, far: 100,
      camera: { rotation: [0, 0, 0, 1], translation: [0, 0, 2], scale: [1, 1, 1] }, model: ident },
    { name: 'default_aspect16_9', fovy: 1, aspect: 1920 / 1080, near: 0.1, far: 100,
      camera: { rotation: [0, 0, 0, 1], translation: [0, 0, 2], scale: [1, 1, 1] }, model: ident },
    { name: 'orbit_yaw30_pitch20', fovy: 1, aspect: 1920 / 1080, near: 0.1, far: 100,
      camera: { rotation: Array.from(quat.multiply(quat.create(), qAxis([0, 1, 0], Math.PI / 6), qAxis([1, 0, 0], -Math.PI / 9))),
                translation: [0.9, 0.7, 1.6], scale: [1, 1, 1] }, model: ident },
    { name: 'scaled_rotated_model', fovy: 0.7, aspect: 4 / 3, near: 0.05, far: 50,
      camera: { rotation: qAxis([0, 1, 0], 0.3), translation: [0.5, 0.1, 2.5], scale: [1, 1, 1] },
      model: { rotation: qAxis([0.267261, 0.534522, 0.801784], 1.1), translation: [0.1, -0.2, 0.05], scale: [1.2, 0.8, 1.5] } },
];
const out = { generator: 'tests/golden/make_mvp_fixture.js', gl_matrix_version: '3.4.1', cases: [] };
for (const c of cases) {
    const r = mvpInverse(c);
    out.cases.push({
        name: c.name, fovy: c.fovy, aspect: c.aspect, near: c.near, far: c.far,
        camera: c.camera, model: c.model,
        forward_bits: bits(r.forward), inverse_bits: bits(r.inverse),
        iso_light: [2, -3, -5], iso_light_bits: bits(isoLight(c, [2, -3, -5])),                 // ISORenderer.js:35-39 default
        iso_light2: [0.25, 1.5, 0.1], iso_light2_bits: bits(isoLight(c, [0.25, 1.5, 0.1])),
    });
}
fs.writeFileSync(path.join(__dirname, 'mvp_inverse.json'), JSON.stringify(out, null, 1));
console.log('wrote', out.cases.length, 'cases');
