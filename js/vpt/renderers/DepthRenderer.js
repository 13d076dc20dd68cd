'use strict';
// src/js/renderers/DepthRenderer.js:13-191 (SURVEY section 8f row 3)
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { native } = require('../native.js');

class DepthRenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_DEPTH; }
static BASE() { return DepthRenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([                                                                  // :17-53
        { name: 'extinction', label: 'Extinction', type: 'spinner', value: 100, min: 0 },
        { name: 'slices', label: 'Slices', type: 'spinner', value: 64, min: 1 },
        { name: 'threshold', label: 'Threshold', type: 'slider', value: 0.1, min: 0, max: 1 },
        { name: 'random', label: 'Random', type: 'checkbox', value: false },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['extinction', 'slices', 'threshold', 'random', 'transferFunction']);   // :55-70
    this._frameNumber = 0;
}

_resetFrame() { native().rendererReset(this._h, null); this._frameNumber = 0; }                // :86-95

_prepareGenerate() {                                                                           // :97-131
    const u = this._newUniforms();
    u.setFloat32(U.STEP, 1 / this.slices, true);
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    u.setFloat32(U.THRESHOLD, this.threshold, true);
    u.setFloat32(U.OFFSET, this.random ? this.rng() : 0, true);
    this._frameNumber++;
    this._u = u;
    return u;
}
_prepareIntegrate() { this._u.setFloat32(U.MIX, 1 / this._frameNumber, true); return this._u; }   // :146
_generateFrame() { this._bindVolume(); native().rendererGenerate(this._h, this._prepareGenerate()); }
_integrateFrame() { native().rendererIntegrate(this._h, this._prepareIntegrate()); }           // :133-149
_renderFrame() { native().rendererRenderFrame(this._h, null); }                                // :151-163
_prepareFused() { this._prepareGenerate(); return this._prepareIntegrate(); }

}
module.exports = { DepthRenderer };
