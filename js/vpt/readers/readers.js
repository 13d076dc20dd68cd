'use strict';
// src/js/readers/{AbstractReader,RAWReader,ZIPReader,BVPReader,ReaderFactory}.js over a loader (js/vpt/loaders).
// Checked against the reference's own readers through tests/golden/readers_r01.json (js/test/test_readers.js).
const { AbstractLoader, BlobLoader } = require('../loaders/loaders.js');

const GL_RED = 6403, GL_R8 = 33321, GL_UNSIGNED_BYTE = 5121;
const GL_RG = 33319, GL_RG8 = 33323;                  // two-channel volumes of BVP manifests
const GL_RGB = 6407, GL_RGB8 = 32849, GL_RGBA = 6408, GL_RGBA8 = 32856;   // byte manifests with more channels (the shaders read .rg)
const GL_FLOAT = 5126, GL_HALF_FLOAT = 5131, GL_R32F = 33326, GL_R16F = 33325;   // float volumes (Volume.js:84-105)

class AbstractReader {                                          // AbstractReader.js:1-15
    constructor(loader) { this._loader = loader; }
    async readMetadata() {}
    async readBlock(block) {}
}

// RAWReader.js:3-70; `loader` may also be the bytes themselves (extension: wrapped in a BlobLoader)
class RAWReader extends AbstractReader {

constructor(loader, options) {
    super(loader instanceof AbstractLoader ? loader : new BlobLoader(loader));
    Object.assign(this, { width: 0, height: 0, depth: 0 }, options || {});
}

async readMetadata() {                                          // :15-63
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

async readBlock(block) {                                        // :65-70
    const sliceBytes = this.width * this.height;
    return await this._loader.readData(block * sliceBytes, (block + 1) * sliceBytes);
}

}

// ZIPReader.js:3-100 — stored entries only (the reference returns an entry's bytes as they lie in the archive)
class ZIPReader extends AbstractReader {

constructor(loader) { super(loader); this._eocd = null; this._cd = null; }

async getFiles() {                                              // :13-19
    if (!this._cd) { await this._readCD(); }
    return this._cd.map(entry => entry.name);
}

async readFile(fileName) {                                      // :21-40
    if (!this._cd) { await this._readCD(); }
    const entry = this._cd.find(e => e.name === fileName);
    if (!entry) { throw new Error(`ZIPReader: file ${fileName} not in CD`); }
    const headerEnd = entry.headerOffset + 30;
    const view = new DataView(await this._loader.readData(entry.headerOffset, headerEnd));
    const dataStart = headerEnd + view.getUint16(26, true) + view.getUint16(28, true);
    return await this._loader.readData(dataStart, dataStart + entry.compressedSize);
}

async _readEOCD() {                                             // :42-58
    const MIN_EOCD_SIZE = 22;
    const length = await this._loader.readLength();
    const offset = Math.max(length - MIN_EOCD_SIZE, 0);
    const view = new DataView(await this._loader.readData(offset, offset + Math.min(length, MIN_EOCD_SIZE)));
    this._eocd = { entries: view.getUint16(10, true), size: view.getUint32(12, true), offset: view.getUint32(16, true) };
}

async _readCD() {                                               // :60-93
    if (!this._eocd) { await this._readEOCD(); }
    const data = await this._loader.readData(this._eocd.offset, this._eocd.offset + this._eocd.size);
    const view = new DataView(data), bytes = new Uint8Array(data);
    let offset = 0;
    const entries = [];
    for (let i = 0; i < this._eocd.entries; i++) {
        const nameLength = view.getUint16(offset + 28, true);
        entries.push({
            gpflag: view.getUint16(offset + 8, true),
            method: view.getUint16(offset + 10, true),
            compressedSize: view.getUint32(offset + 20, true),
            uncompressedSize: view.getUint32(offset + 24, true),
            name: Buffer.from(bytes.subarray(offset + 46, offset + 46 + nameLength)).toString('utf8'),
            headerOffset: view.getUint32(offset + 42, true),
        });
        offset += 46 + nameLength + view.getUint16(offset + 30, true) + view.getUint16(offset + 32, true);
    }
    this._cd = entries;
}

}

// BVPReader.js:4-34
class BVPReader extends AbstractReader {

constructor(loader) { super(loader); this._metadata = null; this._zipReader = new ZIPReader(this._loader); }

async readMetadata() {                                          // :13-20
    const data = await this._zipReader.readFile('manifest.json');
    this._metadata = JSON.parse(Buffer.from(data).toString('utf8'));
    return this._metadata;
}

async readBlock(block) {                                        // :22-29
    if (!this._metadata) { await this.readMetadata(); }
    return await this._zipReader.readFile(this._metadata.blocks[block].url);
}

}

function ReaderFactory(which) {                                 // ReaderFactory.js:5-14
    switch (which) {
        case 'bvp': return BVPReader;
        case 'raw': return RAWReader;
        case 'zip': return ZIPReader;
        default: throw new Error('No suitable class');
    }
}

module.exports = { AbstractReader, RAWReader, ZIPReader, BVPReader, ReaderFactory, GL_RED, GL_R8, GL_RG, GL_RG8, GL_UNSIGNED_BYTE,
    GL_RGB, GL_RGB8, GL_RGBA, GL_RGBA8, GL_FLOAT, GL_HALF_FLOAT, GL_R32F, GL_R16F };
