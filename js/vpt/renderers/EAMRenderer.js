'use strict';
// src/js/renderers/EAMRenderer.js:13-181
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { native } = require('../native.js');

class EAMRenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_EAM; }
static BASE() { return EAMRenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([
        { name: 'extinction', label: 'Extinction', type: 'spinner', value: 100, min: 0 },
        { name: 'slices', label: 'Slices', type: 'spinner', value: 64, min: 1 },
        { name: 'random', label: 'Random', type: 'checkbox', value: true },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['extinction', 'slices', 'random', 'transferFunction']);   // :47-61
    this._frameNumber = 0;
}

_resetFrame() { native().rendererReset(this._h, null); this._frameNumber = 0; }           // :76-86

_prepareGenerate() {                                                                       // :99-116,120
    const u = this._newUniforms();
    u.setFloat32(U.STEP, 1 / this.slices, true);
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    u.setFloat32(U.OFFSET, this.random ? this.rng() : 0, true);
    this._frameNumber++;
    this._u = u;
    return u;
}
_prepareIntegrate() { this._u.setFloat32(U.MIX, 1 / this._frameNumber, true); return this._u; }   // :135
_generateFrame() { this._bindVolume(); native().rendererGenerate(this._h, this._prepareGenerate()); }
_integrateFrame() { native().rendererIntegrate(this._h, this._prepareIntegrate()); }
_renderFrame() { native().rendererRenderFrame(this._h, null); }
_prepareFused() { this._prepareGenerate(); return this._prepareIntegrate(); }

}
module.exports = { EAMRenderer };
