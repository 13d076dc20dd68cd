'use strict';
// src/js/renderers/AbstractRenderer.js:15-157, re-hosted: the frame / accumulation / render buffers are HIP device
// buffers owned by the native renderer; the hooks call the C-ABI through the N-API addon.
//   new R(gl, volume, camera, environmentTexture, { resolution, transform })
//   gl = vpt Context; environmentTexture = { data: Uint8Array RGBA8, width, height } or null (1x1 white).
//   options.resolution: number (square, as in the reference) or { width, height }.
//   options.rng: replaces Math.random() for the per-frame draws (fixed-seed runs).
//   options.shard: { rank, world, rows } — this process renders only its interleaved row blocks (multi-GPU, FrameGather.js).
const { PropertyBag } = require('../PropertyBag.js');
const { Transform, mvpInverseMatrix } = require('../scene.js');
const { native } = require('../native.js');

// byte layout of struct vpt_uniforms (include/vpt.h)
const U = { MVP: 0, SEED: 64, OFFSET: 68, STEP: 72, EXTINCTION: 76, ANISOTROPY: 80, BOUNCES: 84, STEPS: 88, LIGHT: 92, MIX: 104, BLUR: 108,
            ISOVALUE: 112, GRADIENT_STEP: 116, THRESHOLD: 120, SIZE: 128 };

const GL = { NEAREST: 9728, CLAMP_TO_EDGE: 33071, RED: 6403, RG: 33319, RGBA: 6408, R8: 33321, R32F: 33326, RG32F: 33328,
    RGBA16F: 34842, RGBA32F: 34836, UNSIGNED_BYTE: 5121, FLOAT: 5126 };
const KIND_NAMES = ['mip', 'eam', 'mcs', 'mcm', 'iso', 'depth', 'lao', 'dos'];      // VPT_RENDERER_* (include/vpt.h)
const U8 = ['RGBA', 'RGBA', 'UNSIGNED_BYTE'], F4 = ['RGBA', 'RGBA32F', 'FLOAT'];
const BUFFER_FORMATS = {            // renderer name -> [frame attachments, accumulation attachments] as [format, iformat, type]
    mip: [[['RED', 'R8', 'UNSIGNED_BYTE']], [['RED', 'R8', 'UNSIGNED_BYTE']]],              // MIPRenderer.js:133-157
    eam: [[U8], [U8]], lao: [[U8], [U8]],                                                   // EAMRenderer.js:155-179
    mcs: [[F4], [F4]],                                                                      // MCSRenderer.js:156-180
    mcm: [[F4], [F4, F4, F4, F4]],                                                          // MCMRenderer.js:201-263
    iso: [[['RGBA', 'RGBA16F', 'FLOAT']], [['RGBA', 'RGBA16F', 'FLOAT']]],
    depth: [[['RED', 'R32F', 'FLOAT']], [['RED', 'R32F', 'FLOAT']]],
    dos: [[F4], [F4, ['RED', 'R32F', 'FLOAT']]],                                            // DOSRenderer.js:277-305
};

class AbstractRenderer extends PropertyBag {

constructor(gl, volume, camera, environmentTexture, options) {
    super();
    options = options || {};
    this._resolution = options.resolution !== undefined ? options.resolution : 512;       // AbstractRenderer.js:20
    this._gl = gl;
    this._volume = volume;
    this._camera = camera;
    this._environmentTexture = environmentTexture;
    this._volumeTransform = options.transform !== undefined ? options.transform : new Transform();   // :27
    this.rng = options.rng || Math.random;
    this.fused = options.fused !== undefined ? options.fused : true;
    this._h = null;
    this._shard = options.shard || null;
    this._boundVolume = undefined;
    this._rebuildBuffers();
    if (environmentTexture) {
        native().rendererSetEnvironment(this._h, environmentTexture.data, environmentTexture.width, environmentTexture.height);
    }
}

_size() {
    const r = this._resolution;
    return typeof r === 'number' ? [r, r] : [r.width, r.height];
}

_rebuildBuffers() {                                                                          // :78-92
    const N = native();
    const size = this._size();
    if (!this._h) {
        this._h = N.rendererCreate(this._gl._h, this.constructor.KIND(), size[0], size[1]);
        if (this._shard) { N.rendererSetShard(this._h, this._shard.rank, this._shard.world, this._shard.rows || 8); }
    } else {
        N.rendererResize(this._h, size[0], size[1]);
    }
}

_bindVolume() {
    const tex = this._volume ? this._volume.getTexture() : null;
    if (tex !== this._boundVolume) { native().rendererSetVolume(this._h, tex); this._boundVolume = tex; }
}

_newUniforms() {
    const buf = new ArrayBuffer(U.SIZE);
    new Float32Array(buf, 0, 16).set(mvpInverseMatrix(this._camera, this._volumeTransform));
    return new DataView(buf);
}

destroy() { if (this._h) { native().rendererDestroy(this._h); this._h = null; } }           // :51-58

render() {                                                                                   // :60-70
    if (this.fused && !this._hooksOverridden()) { this._renderFused(); return; }
    this._generateFrame();
    this._integrateFrame();
    this._renderFrame();
}

reset() { this._resetFrame(); }                                                              // :72-76

setVolume(volume) { this._volume = volume; this.reset(); }                                   // :94-97

// :99-104 — { data: Uint8Array RGBA8, width, height } (the reference passes a TexImageSource)
setTransferFunction(transferFunction) {
    native().rendererSetTransferFunction(this._h, transferFunction.data, transferFunction.width, transferFunction.height);
}

// the reference re-fills the context-owned environment texture in place (RenderingContext.js:135-140); here the
// renderer holds a device copy, so the context hands the new image down
setEnvironmentMap(image) {
    this._environmentTexture = image;
    native().rendererSetEnvironment(this._h, image.data, image.width, image.height);
}

setResolution(resolution) {                                                                  // :106-112
    if (resolution !== this._resolution) {
        this._resolution = resolution;
        this._rebuildBuffers();
        this.reset();
    }
}

// :114-116 — the RGBA16F colour attachment, read back as raw half bits [rows][width][4]
getTexture() {
    const size = this._size();
    const rows = native().rendererLocalRows(this._h);
    const out = new Uint16Array(rows * size[0] * 4);
    native().rendererRead(this._h, native().VPT_BUFFER_RENDER, out);
    return { data: out, width: size[0], height: rows, format: 'RGBA16F' };
}

// extension: `count` render() passes by one native call.  mode: native().VPT_PLAY_EAGER | _GRAPH | _FUSED (one launch,
// state / accumulator in registers between passes; not LAO / DOS) | _FRAMES (MCM: _FUSED that writes every frame, readFrameSlot).  The per-frame draws are taken exactly as render() would.
play(count, mode) {
    this._bindVolume();
    const vars = new Float32Array(8 * count);
    let u = null;
    for (let k = 0; k < count; k++) {
        u = this._prepareFused();
        vars[8 * k] = u.getFloat32(U.SEED, true); vars[8 * k + 1] = u.getFloat32(U.OFFSET, true); vars[8 * k + 2] = u.getFloat32(U.MIX, true);
        for (let i = 0; i < 3; i++) { vars[8 * k + 4 + i] = u.getFloat32(U.LIGHT + 4 * i, true); }
    }
    native().rendererPlay(this._h, u, vars, mode === undefined ? native().VPT_PLAY_EAGER : mode);
}

read(buffer, out) { native().rendererRead(this._h, buffer, out); return out; }
// frame `slot` of the last play(count, VPT_PLAY_FRAMES) call (MCM): the image the slot-th of `count` render() calls would have shown
readFrameSlot(slot) {
    const size = this._size();
    const rows = native().rendererLocalRows(this._h);
    const out = new Uint16Array(rows * size[0] * 4);
    native().rendererReadFrameSlot(this._h, slot, out);
    return { data: out, width: size[0], height: rows, format: 'RGBA16F' };
}
sampleCount() { return native().rendererSampleCount(this._h); }

_hooksOverridden() {
    const base = this.constructor.BASE().prototype, mine = Object.getPrototypeOf(this);
    return ['_generateFrame', '_integrateFrame', '_renderFrame'].some(n => mine[n] !== base[n]);
}

_resetFrame() {}
_generateFrame() {}
_integrateFrame() {}
_renderFrame() {}

// Buffer specs (AbstractRenderer.js:134-155 and the subclasses' _get*BufferSpec).  In the reference these lists drive the allocation of
// the WebGL attachments (_rebuildBuffers, :78-92); here the native renderer owns the HIP buffers and the hooks DESCRIBE them with the
// reference's GL enums, one object per attachment in attachment order.
_spec(fmt) {
    const size = this._size();
    return { width: size[0], height: size[1], min: GL.NEAREST, mag: GL.NEAREST, format: GL[fmt[0]], iformat: GL[fmt[1]], type: GL[fmt[2]] };
}
_getFrameBufferSpec() { return BUFFER_FORMATS[KIND_NAMES[this.constructor.KIND()]][0].map(f => this._spec(f)); }
_getAccumulationBufferSpec() { return BUFFER_FORMATS[KIND_NAMES[this.constructor.KIND()]][1].map(f => this._spec(f)); }
_getRenderBufferSpec() {                                                                     // AbstractRenderer.js:142-155
    const d = this._spec(['RGBA', 'RGBA16F', 'FLOAT']);
    d.wrapS = d.wrapT = GL.CLAMP_TO_EDGE;
    return [d];
}
// the uniforms of one whole render() pass, with the per-frame draws taken in hook order (subclasses)
_prepareFused() { return null; }
_renderFused() { this._bindVolume(); native().rendererRender(this._h, this._prepareFused()); }

}

function installChangeHandler(renderer, resetOn) {
    renderer.addEventListener('change', e => {
        const name = e.detail.name;
        if (name === 'transferFunction') { renderer.setTransferFunction(renderer.transferFunction); }
        if (resetOn.indexOf(name) >= 0) { renderer.reset(); }
    });
}
const transferFunctionProperty = () => ({
    name: 'transferFunction', label: 'Transfer function', type: 'transfer-function', value: new Uint8Array(256),
});

module.exports = { AbstractRenderer, U, installChangeHandler, transferFunctionProperty };
