'use strict';
// src/js/renderers/MIPRenderer.js:13-159
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { native } = require('../native.js');

class MIPRenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_MIP; }
static BASE() { return MIPRenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([
        { name: 'steps', label: 'Steps', type: 'spinner', value: 64, min: 1 },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['transferFunction']);                   // :34-46 — 'steps' does not reset
}

_resetFrame() { native().rendererReset(this._h, null); }               // :61-68

_prepareGenerate() {                                                    // :82-97
    const u = this._newUniforms();
    u.setFloat32(U.STEP, 1 / this.steps, true);
    u.setFloat32(U.OFFSET, this.rng(), true);
    this._u = u;
    return u;
}
_generateFrame() { this._bindVolume(); native().rendererGenerate(this._h, this._prepareGenerate()); }
_integrateFrame() { native().rendererIntegrate(this._h, this._u); }    // :102-117
_renderFrame() { native().rendererRenderFrame(this._h, null); }        // :119-131
_prepareFused() { return this._prepareGenerate(); }

}
module.exports = { MIPRenderer };
