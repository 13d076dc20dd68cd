'use strict';
// src/js/renderers/DOSRenderer.js:14-316 (SURVEY section 8f row 3): directional occlusion shading, `steps` view-aligned
// slices per render() call until the sweep has passed the far corner of the volume
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { PerspectiveCamera } = require('../scene.js');
const { vec3, mat4 } = require('../math.js');
const { native } = require('../native.js');

class DOSRenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_DOS; }
static BASE() { return DOSRenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([                                                                  // :18-63
        { name: 'steps', label: 'Steps', type: 'spinner', value: 50, min: 1 },
        { name: 'slices', label: 'Slices', type: 'spinner', value: 200, min: 1 },
        { name: 'extinction', label: 'Extinction', type: 'spinner', value: 100, min: 0 },
        { name: 'aperture', label: 'Aperture', type: 'spinner', value: 30, min: 0, max: 89 },
        { name: 'samples', label: 'Samples', type: 'spinner', value: 8, min: 1, max: 200, step: 1 },
        transferFunctionProperty(),
    ]);
    this.addEventListener('change', e => {                                                     // :72-74, ahead of the reset
        if (e.detail.name === 'samples') { this.generateOcclusionSamples(); }
    });
    installChangeHandler(this, ['slices', 'extinction', 'aperture', 'samples', 'transferFunction']);   // :76-84
    this._depth = 0;
    this._minDepth = 0;
    this._maxDepth = 0;
    this.fused = false;                                  // there is no single-launch form of this renderer
    this.generateOcclusionSamples();
}

generateOcclusionSamples() {                                                                   // :103-140
    const data = new Float32Array(this.samples * 2);
    let averagex = 0;
    let averagey = 0;
    for (let i = 0; i < this.samples; i++) {
        const r = Math.sqrt(this.rng());
        const phi = this.rng() * 2 * Math.PI;
        const x = r * Math.cos(phi);
        const y = r * Math.sin(phi);
        averagex += x / this.samples;
        averagey += y / this.samples;
        data[2 * i + 0] = x;
        data[2 * i + 1] = y;
    }
    for (let i = 0; i < this.samples; i++) {
        data[2 * i + 0] -= averagex;
        data[2 * i + 1] -= averagey;
    }
    this._occlusionSamples = data;
    native().rendererSetOcclusionSamples(this._h, data);
}

calculateDepth() {                                                                             // :142-167
    const centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]);
    const modelMatrix = this._volumeTransform.globalMatrix;
    const viewMatrix = this._camera.transform.inverseGlobalMatrix;
    const matrix = mat4.create();
    mat4.multiply(matrix, centerMatrix, matrix);
    mat4.multiply(matrix, modelMatrix, matrix);
    mat4.multiply(matrix, viewMatrix, matrix);
    const corners = [
        [0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1],
        [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1],
    ];
    const depths = corners.map(v => -vec3.transformMat4(v, v, matrix)[2]);
    return [Math.min(...depths), Math.max(...depths)];
}

_resetFrame() {                                                                                // :169-185
    [this._minDepth, this._maxDepth] = this.calculateDepth();
    this._minDepth = Math.max(this._minDepth, 0);
    this._depth = this._minDepth;
    native().rendererReset(this._h, null);
}

// the uniforms of :212-236 and, per pass of the loop :240-259, (uOcclusionScale.x, uOcclusionScale.y, uDepth)
_prepareSlices() {
    const u = this._newUniforms();
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    const sliceDistance = (this._maxDepth - this._minDepth) / this.slices;
    u.setFloat32(U.STEP, sliceDistance, true);
    const projectionMatrix = this._camera.getComponent(PerspectiveCamera).projectionMatrix;
    const rows = [];
    for (let step = 0; step < this.steps; step++) {
        if (this._depth > this._maxDepth) { break; }
        const correction = [1, 1, -this._depth];
        vec3.transformMat4(correction, correction, projectionMatrix);
        const occlusionExtent = sliceDistance * Math.tan(this.aperture * Math.PI / 180);
        correction[0] *= occlusionExtent;
        correction[1] *= occlusionExtent;
        rows.push(correction[0], correction[1], correction[2]);
        this._depth += sliceDistance;
    }
    this._u = u;
    this._slices = new Float32Array(rows);
    return u;
}
_generateFrame() {}                                                                            // AbstractRenderer.js:122-124
_integrateFrame() {                                                                            // :187-262
    this._bindVolume();
    const u = this._prepareSlices();
    native().rendererIntegrateSlices(this._h, u, this._slices);
}
_renderFrame() { native().rendererRenderFrame(this._h, null); }                                // :264-277
play() { throw new Error('frame sequences are not defined for the DOS renderer: drive it slice by slice'); }

}
module.exports = { DOSRenderer };
