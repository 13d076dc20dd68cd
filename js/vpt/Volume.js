'use strict';
// src/js/Volume.js:3-127 on HIP device memory; fed by a reader (js/vpt/readers/readers.js).
const { EventTarget, CustomEvent } = require('./EventTarget.js');
const { native } = require('./native.js');

const { RAWReader, GL_RED, GL_RG, GL_UNSIGNED_BYTE } = require('./readers/readers.js');

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
    if (modality.type !== GL_UNSIGNED_BYTE || (modality.format !== GL_RED && modality.format !== GL_RG)) {
        throw new Error('Unknown volume datatype: ' + modality.type);                           // Volume.js:103
    }
    const { width, height, depth } = modality.dimensions;
    this.texture = N.volumeCreate(this._gl._h, width, height, depth, modality.format === GL_RG ? N.VPT_FORMAT_RG8 : N.VPT_FORMAT_R8);
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
