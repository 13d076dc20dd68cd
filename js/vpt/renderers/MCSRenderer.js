'use strict';
// src/js/renderers/MCSRenderer.js:13-182
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { native } = require('../native.js');

class MCSRenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_MCS; }
static BASE() { return MCSRenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([
        { name: 'extinction', label: 'Extinction', type: 'spinner', value: 1, min: 0 },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['extinction', 'transferFunction']);     // :34-47
    this._frameNumber = 1;
}

_resetFrame() { native().rendererReset(this._h, null); this._frameNumber = 1; }            // :62-72

_prepareGenerate() {                                                                        // :88-117
    const u = this._newUniforms();
    u.setFloat32(U.SEED, this.rng(), true);
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    let x, y, z, length;
    do {
        x = this.rng() * 2 - 1;
        y = this.rng() * 2 - 1;
        z = this.rng() * 2 - 1;
        length = Math.sqrt(x * x + y * y + z * z);
    } while (length > 1);
    u.setFloat32(U.LIGHT, x / length, true);
    u.setFloat32(U.LIGHT + 4, y / length, true);
    u.setFloat32(U.LIGHT + 8, z / length, true);
    this._u = u;
    return u;
}
_prepareIntegrate() {                                                                       // :137-139
    this._u.setFloat32(U.MIX, 1 / this._frameNumber, true);
    this._frameNumber += 1;
    return this._u;
}
_generateFrame() { this._bindVolume(); native().rendererGenerate(this._h, this._prepareGenerate()); }
_integrateFrame() { native().rendererIntegrate(this._h, this._prepareIntegrate()); }
_renderFrame() { native().rendererRenderFrame(this._h, null); }
_prepareFused() { this._prepareGenerate(); return this._prepareIntegrate(); }

}
module.exports = { MCSRenderer };
