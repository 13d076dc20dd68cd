'use strict';
// src/js/renderers/LAORenderer.js:13-245 (SURVEY section 8f row 3)
const { AbstractRenderer, U, installChangeHandler, transferFunctionProperty } = require('./AbstractRenderer.js');
const { native } = require('../native.js');

class LAORenderer extends AbstractRenderer {

static KIND() { return native().VPT_RENDERER_LAO; }
static BASE() { return LAORenderer; }

constructor(gl, volume, camera, environmentTexture, options) {
    super(gl, volume, camera, environmentTexture, options);
    this.registerProperties([                                                                  // :17-108
        { name: 'extinction', label: 'Extinction', type: 'spinner', value: 100, min: 0 },
        { name: 'localAmbientOcclusion', label: 'Local Ambient Occlusion', type: 'checkbox', value: true },
        { name: 'LAOWeight', label: 'LAO Weight', type: 'spinner', value: 0.69, min: 0, max: 1 },
        { name: 'numLAOSamples', label: '# of LAO Samples', type: 'spinner', value: 1, min: 1 },
        { name: 'LAOStepSize', label: 'LAO Stem Size', type: 'spinner', value: 0.05, min: 0 },
        { name: 'softShadows', label: 'Soft Shadows', type: 'checkbox', value: true },
        { name: 'shadowsWeight', label: 'Shadows Weight', type: 'spinner', value: 0.54, min: 0, max: 1 },
        { name: 'numShadowSamples', label: '# of Shadow Samples', type: 'spinner', value: 10, min: 1 },
        { name: 'lightRadious', label: 'Light Radious', type: 'spinner', value: 0.19, min: 0 },
        { name: 'lightPosition', label: 'Light position', type: 'vector-spinner', value: [2, 12, 3] },
        { name: 'lightCoeficient', label: 'Light Coeficient', type: 'spinner', value: 1.0, min: 0 },
        { name: 'slices', label: 'Slices', type: 'spinner', value: 64, min: 1 },
        transferFunctionProperty(),
    ]);
    installChangeHandler(this, ['extinction', 'slices', 'transferFunction']);                  // :110-124
}

// the gl.uniform* block of :159-169 as struct vpt_lao_params (include/vpt.h)
laoParams() {
    const p = new DataView(new ArrayBuffer(48));
    p.setInt32(0, this.localAmbientOcclusion ? 1 : 0, true);
    p.setFloat32(4, this.LAOWeight, true);
    p.setInt32(8, this.numLAOSamples, true);
    p.setFloat32(12, this.LAOStepSize, true);
    p.setInt32(16, this.softShadows ? 1 : 0, true);
    p.setFloat32(20, this.shadowsWeight, true);
    p.setInt32(24, this.numShadowSamples, true);
    p.setFloat32(28, this.lightRadious, true);
    p.setFloat32(32, this.lightCoeficient, true);
    for (let i = 0; i < 3; i++) p.setFloat32(36 + 4 * i, this.lightPosition[i], true);
    return p;
}

_resetFrame() { native().rendererReset(this._h, null); }                                       // :138-145

_prepareGenerate() {                                                                           // :147-186
    const u = this._newUniforms();
    u.setFloat32(U.STEP, 1 / this.slices, true);
    u.setFloat32(U.EXTINCTION, this.extinction, true);
    u.setFloat32(U.OFFSET, this.rng(), true);            // :170 draws Math.random() although the shader never reads uOffset
    native().rendererSetLaoParams(this._h, this.laoParams());
    this._u = u;
    return u;
}
_generateFrame() { this._bindVolume(); native().rendererGenerate(this._h, this._prepareGenerate()); }
_integrateFrame() { native().rendererIntegrate(this._h, this._u); }                            // :188-203
_renderFrame() { native().rendererRenderFrame(this._h, null); }                                // :205-217
_prepareFused() { return this._prepareGenerate(); }

}
module.exports = { LAORenderer };
