'use strict';
// src/js/tonemappers/AbstractToneMapper.js:10-81, re-hosted over the C-ABI (vpt_tonemapper_*).
//   new T(gl, texture, { resolution })
//   gl = vpt Context.  texture = what the reference gets from renderer.getTexture(): here the renderer itself (its
//   RGBA16F render buffer is read in place in HBM), { data: Uint16Array RGBA16F bits, width, height } (uploaded), or
//   null (the 1x1 white placeholder of RenderingContext.js:176-181).
const { PropertyBag } = require('../PropertyBag.js');
const { native } = require('../native.js');

// index of each field of struct vpt_tonemap_params in its Float32Array(8)
const P = { LOW: 0, MID: 1, HIGH: 2, SATURATION: 3, MIN: 4, MAX: 5, EXPOSURE: 6, GAMMA: 7 };

class AbstractToneMapper extends PropertyBag {

constructor(gl, texture, options) {
    super();
    options = options || {};
    this._resolution = options.resolution !== undefined ? options.resolution : 512;          // :15
    this._gl = gl;
    this._h = null;
    this._rebuildBuffers();
    this.setTexture(texture);
}

_size() {
    const r = this._resolution;
    return typeof r === 'number' ? [r, r] : [r.width, r.height];
}

destroy() { if (this._h) { native().tonemapperDestroy(this._h); this._h = null; } }         // :28-33

render() { this._renderFrame(); }                                                           // :35-38

setTexture(texture) {                                                                       // :40-42
    const N = native();
    this._texture = texture;
    if (!texture) { N.tonemapperSetSource(this._h, null); }
    else if (texture._h && typeof texture.render === 'function') { N.tonemapperSetSource(this._h, texture._h); }
    else { N.tonemapperSetSourceImage(this._h, texture.data, texture.width, texture.height); }
}

// :44-46 — the RGBA8 colour attachment, read back
getTexture() {
    const N = native();
    const rows = N.tonemapperRows(this._h), w = this._size()[0];
    const out = new Uint8Array(rows * w * 4);
    N.tonemapperRead(this._h, out);
    return { data: out, width: w, height: rows, format: 'RGBA8' };
}

_rebuildBuffers() {                                                                         // :48-54
    const N = native(), size = this._size();
    if (!this._h) { this._h = N.tonemapperCreate(this._gl._h, this.constructor.KIND(), size[0], size[1]); }
    else { N.tonemapperResize(this._h, size[0], size[1]); }
}

setResolution(resolution) {                                                                 // :56-61
    if (resolution !== this._resolution) {
        this._resolution = resolution;
        this._rebuildBuffers();
    }
}

_params() { return new Float32Array([0, 0.5, 1, 1, 0, 1, 1, 2.2]); }

_renderFrame() { native().tonemapperRender(this._h, this._params()); }                      // :63-65

}

// the eight curve mappers share one host body (e.g. src/js/tonemappers/ReinhardToneMapper.js:10-59)
class ExposureGammaToneMapper extends AbstractToneMapper {

constructor(gl, texture, options) {
    super(gl, texture, options);
    this.registerProperties([                                                               // :14-29
        { name: 'exposure', label: 'Exposure', type: 'spinner', value: 1, min: 0 },
        { name: 'gamma', label: 'Gamma', type: 'spinner', value: 2.2, min: 0 },
    ]);
}

_params() {                                                                                 // :53-54
    const p = super._params();
    p[P.EXPOSURE] = this.exposure; p[P.GAMMA] = this.gamma;
    return p;
}

}

module.exports = { AbstractToneMapper, ExposureGammaToneMapper, P };
