'use strict';
// src/js/renderers/DOSRenderer.js:14-316 (SURVEY section 8f row 3): directional occlusion shading, `steps` view-aligned
// slices per render() call until the sweep has passed the far corner of the volume
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { PerspectiveCamera } = require('../scene.js');
const { occlusionTaps, viewDepthRange, sliceTriples } = require('./dosSweep.js');
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
    this._occlusionSamples = occlusionTaps(this.rng, this.samples);
    native().rendererSetOcclusionSamples(this._h, this._occlusionSamples);
}

calculateDepth() {                                                                             // :142-167
    return viewDepthRange(this._volumeTransform.globalMatrix, this._camera.transform.inverseGlobalMatrix);
}

_resetFrame() {                                                                                // :169-185
    const [nearest, farthest] = this.calculateDepth();
    this._minDepth = Math.max(nearest, 0);
    this._maxDepth = farthest;
    this._depth = this._minDepth;
    native().rendererReset(this._h, null);
}

// the uniforms of :212-236 and, per pass of the loop :240-259, (uOcclusionScale.x, uOcclusionScale.y, uDepth)
_prepareSlices() {
    const u = this._newUniforms();
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    const sliceDistance = (this._maxDepth - this._minDepth) / this.slices;
    u.setFloat32(U.STEP, sliceDistance, true);
    const sweep = { depth: this._depth, farthest: this._maxDepth };
    this._slices = sliceTriples(sweep, this.steps, sliceDistance, this.aperture, this._camera.getComponent(PerspectiveCamera).projectionMatrix);
    this._depth = sweep.depth;
    this._u = u;
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
