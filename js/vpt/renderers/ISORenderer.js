'use strict';
// src/js/renderers/ISORenderer.js:13-199 (SURVEY section 8f row 3)
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { isoLightDirection } = require('../scene.js');
const { native } = require('../native.js');

class ISORenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_ISO; }
static BASE() { return ISORenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([                                                                  // :17-46
        { name: 'steps', label: 'Steps', type: 'spinner', value: 50, min: 1 },
        { name: 'isovalue', label: 'Isovalue', type: 'slider', value: 0.5, min: 0, max: 1 },
        { name: 'light', label: 'Light direction', type: 'vector-spinner', value: [2, -3, -5] },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['isovalue', 'transferFunction']);                             // :48-61 — steps and light do not reset
}

_resetFrame() { native().rendererReset(this._h, null); }                                       // :75-82

_prepareGenerate() {                                                                           // :84-116
    const u = this._newUniforms();
    u.setUint32(U.STEPS, this.steps, true);
    u.setFloat32(U.STEP, Math.fround(1.0) / Math.fround(this.steps), true);                     // the shader's 1.0 / float(uSteps), ISORenderer.glsl:64
    u.setFloat32(U.OFFSET, this.rng(), true);
    u.setFloat32(U.ISOVALUE, this.isovalue, true);
    this._u = u;
    return u;
}
_prepareRender() {                                                                             // :136-171
    const light = isoLightDirection(this._camera, this._volumeTransform, this.light);
    for (let i = 0; i < 3; i++) { this._u.setFloat32(U.LIGHT + 4 * i, light[i], true); }
    this._u.setFloat32(U.GRADIENT_STEP, 0.005, true);                                           // :168
    return this._u;
}
_generateFrame() { this._bindVolume(); native().rendererGenerate(this._h, this._prepareGenerate()); }
_integrateFrame() { native().rendererIntegrate(this._h, this._u); }                            // :118-134
_renderFrame() { this._bindVolume(); native().rendererRenderFrame(this._h, this._prepareRender()); }
_prepareFused() { this._prepareGenerate(); return this._prepareRender(); }

}
module.exports = { ISORenderer };
