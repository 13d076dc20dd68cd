'use strict';
// src/js/renderers/MCMRenderer.js:13-265
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { native } = require('../native.js');

class MCMRenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_MCM; }
static BASE() { return MCMRenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([
        { name: 'extinction', label: 'Extinction', type: 'spinner', value: 1, min: 0 },
        { name: 'anisotropy', label: 'Anisotropy', type: 'slider', value: 0, min: -1, max: 1 },
        { name: 'bounces', label: 'Max bounces', type: 'spinner', value: 8, min: 0 },
        { name: 'steps', label: 'Steps', type: 'spinner', value: 8, min: 0 },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['extinction', 'anisotropy', 'bounces', 'transferFunction']);   // :56-71 — not 'steps'
}

_resetFrame() {                                                          // :85-116
    const u = this._newUniforms();
    u.setFloat32(U.SEED, this.rng(), true);
    u.setFloat32(U.BLUR, 0, true);
    native().rendererReset(this._h, u);
}

_generateFrame() {}                                                      // :118-119 (empty)

_prepareIntegrate() {                                                    // :155-175
    const u = this._newUniforms();
    u.setFloat32(U.SEED, this.rng(), true);
    u.setFloat32(U.BLUR, 0, true);
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    u.setFloat32(U.ANISOTROPY, this.anisotropy, true);
    u.setUint32(U.BOUNCES, this.bounces, true);
    u.setUint32(U.STEPS, this.steps, true);
    this._u = u;
    return u;
}
_integrateFrame() { this._bindVolume(); native().rendererIntegrate(this._h, this._prepareIntegrate()); }   // :121-185
_renderFrame() { native().rendererRenderFrame(this._h, null); }                                              // :187-199
_prepareFused() { return this._prepareIntegrate(); }

}
module.exports = { MCMRenderer };
