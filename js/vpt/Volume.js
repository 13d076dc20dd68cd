'use strict';
// src/js/Volume.js:3-127 on HIP device memory; fed by a reader (js/vpt/readers/readers.js).
const { EventTarget, CustomEvent } = require('./EventTarget.js');
const { native } = require('./native.js');

const { RAWReader, GL_RED, GL_RG, GL_RGB, GL_RGBA, GL_UNSIGNED_BYTE, GL_FLOAT, GL_HALF_FLOAT } = require('./readers/readers.js');

// IEEE half -> float (exact)
function halfToFloat(h) {
    const s = (h & 0x8000) ? -1 : 1, e = (h >> 10) & 31, m = h & 1023;
    if (e === 0) { return s * m * Math.pow(2, -24); }
    if (e === 31) { return m ? NaN : s * Infinity; }
    return s * (1 + m / 1024) * Math.pow(2, e - 15);
}

// (native format, channels in the file, element kind) for a manifest's (format, type): what a WebGL2 sampler3D can filter —
// UNSIGNED_BYTE with 1-4 channels (the shaders read .rg: further channels are dropped on upload), FLOAT / HALF_FLOAT with one
// channel (R32F / R16F).  Anything else raises the reference's error (Volume.js:103).
function deviceFormat(N, modality) {
    const t = modality.type, f = modality.format;
    if (t === GL_UNSIGNED_BYTE && (f === GL_RED || f === GL_RG || f === GL_RGB || f === GL_RGBA)) {
        const n = f === GL_RED ? 1 : (f === GL_RG ? 2 : (f === GL_RGB ? 3 : 4));
        return { fmt: n === 1 ? N.VPT_FORMAT_R8 : N.VPT_FORMAT_RG8, channels: n, kind: 'u8' };
    }
    if ((t === GL_FLOAT || t === GL_HALF_FLOAT) && (f === GL_RED || f === GL_RG || f === GL_RGB || f === GL_RGBA)) {
        const n = f === GL_RED ? 1 : (f === GL_RG ? 2 : (f === GL_RGB ? 3 : 4));
        return { fmt: n === 1 ? N.VPT_FORMAT_R32F : N.VPT_FORMAT_RG32F, channels: n, kind: t === GL_FLOAT ? 'f32' : 'f16' };
    }
    throw new Error('Unknown volume datatype: ' + t);
}

// a block as the bytes vpt_volume_upload_block takes: u8 with at most two channels, or float32
function blockBytes(data, df) {
    const u8 = data instanceof Uint8Array ? data : new Uint8Array(data.buffer || data, data.byteOffset || 0, data.byteLength);
    if (df.kind === 'u8') {
        if (df.channels <= 2) { return u8; }
        const n = u8.length / df.channels, out = new Uint8Array(2 * n);
        for (let i = 0; i < n; i++) { out[2 * i] = u8[df.channels * i]; out[2 * i + 1] = u8[df.channels * i + 1]; }
        return out;
    }
    // float texels: float32 (half widened exactly), at most two channels (the shaders read .rg)
    let f;
    if (df.kind === 'f32') {
        f = new Float32Array(u8.buffer.slice(u8.byteOffset, u8.byteOffset + u8.byteLength));
    } else {
        const h = new Uint16Array(u8.buffer.slice(u8.byteOffset, u8.byteOffset + u8.byteLength));
        f = new Float32Array(h.length);
        for (let i = 0; i < h.length; i++) { f[i] = halfToFloat(h[i]); }
    }
    if (df.channels > 2) {
        const n = f.length / df.channels, out = new Float32Array(2 * n);
        for (let i = 0; i < n; i++) { out[2 * i] = f[df.channels * i]; out[2 * i + 1] = f[df.channels * i + 1]; }
        f = out;
    }
    return new Uint8Array(f.buffer);
}

class Volume extends EventTarget {

constructor(gl, reader, options) {
    super();
    this._gl = gl;
    this._reader = reader;
    this.metadata = null;
    this.ready = false;
    this.texture = null;
    this.modality = null;
}

destroy() {
    if (this.texture) { native().volumeDestroy(this.texture); this.texture = null; this.ready = false; }
}

async readMetadata() {
    if (!this.metadata) { this.metadata = await this._reader.readMetadata(); }
    return this.metadata;
}

async readModality(modalityName) {
    const N = native();
    this.ready = false;
    if (!this.metadata) { await this.readMetadata(); }
    const modality = this.metadata.modalities.find(m => m.name === modalityName);
    if (!modality) { throw new Error(`Modality '${modalityName}' does not exist`); }          // Volume.js:40
    this.modality = modality;
    if (this.texture) { N.volumeDestroy(this.texture); this.texture = null; }
    const df = deviceFormat(N, modality);                                                       // Volume.js:58-60,84-105
    const { width, height, depth } = modality.dimensions;
    this.texture = N.volumeCreate(this._gl._h, width, height, depth, df.fmt);
    for (const { index, position } of modality.placements) {
        const data = await this._reader.readBlock(index);
        const d = this.metadata.blocks[index].dimensions;
        N.volumeUploadBlock(this.texture, position.x, position.y, position.z, d.width, d.height, d.depth, blockBytes(data, df));
        const progress = (index + 1) / modality.placements.length;
        this.dispatchEvent(new CustomEvent('progress', { detail: progress }));
    }
    N.volumeFinalize(this.texture);
    this.ready = true;
}

async load() { await this.readModality('default'); }

getTexture() { return this.ready ? this.texture : null; }

setFilter(filter) {
    if (!this.texture) { return; }
    const N = native();
    N.volumeSetFilter(this.texture, filter === 'linear' ? N.VPT_FILTER_LINEAR : N.VPT_FILTER_NEAREST);
}

}
module.exports = { Volume, RAWReader };
