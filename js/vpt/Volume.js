'use strict';
// src/js/Volume.js:3-127 on HIP device memory, and the in-memory form of src/js/readers/RAWReader.js:3-70.
const { EventTarget, CustomEvent } = require('./EventTarget.js');
const { native } = require('./native.js');

const GL_RED = 6403, GL_R8 = 33321, GL_UNSIGNED_BYTE = 5121;

class RAWReader {

constructor(data, options) {
    options = options || {};
    this.width = options.width || 0;
    this.height = options.height || 0;
    this.depth = options.depth || 0;
    this._data = data instanceof Uint8Array ? data : new Uint8Array(data);
}

async readMetadata() {
    const metadata = {
        meta: { version: 1 },
        modalities: [{
            name: 'default',
            dimensions: { width: this.width, height: this.height, depth: this.depth },
            transform: { matrix: [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1] },
            format: GL_RED, internalFormat: GL_R8, type: GL_UNSIGNED_BYTE,
            placements: [],
        }],
        blocks: [],
    };
    for (let i = 0; i < this.depth; i++) {
        metadata.modalities[0].placements.push({ index: i, position: { x: 0, y: 0, z: i } });
        metadata.blocks.push({ url: 'default', format: 'raw', dimensions: { width: this.width, height: this.height, depth: 1 } });
    }
    return metadata;
}

async readBlock(block) {
    const sliceBytes = this.width * this.height;
    return this._data.subarray(block * sliceBytes, (block + 1) * sliceBytes);
}

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
    if (modality.type !== GL_UNSIGNED_BYTE || modality.format !== GL_RED) {
        throw new Error('Unknown volume datatype: ' + modality.type);                           // Volume.js:103
    }
    const { width, height, depth } = modality.dimensions;
    this.texture = N.volumeCreate(this._gl._h, width, height, depth, N.VPT_FORMAT_R8);
    for (const { index, position } of modality.placements) {
        const data = await this._reader.readBlock(index);
        const d = this.metadata.blocks[index].dimensions;
        N.volumeUploadBlock(this.texture, position.x, position.y, position.z, d.width, d.height, d.depth,
            data instanceof Uint8Array ? data : new Uint8Array(data));
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
