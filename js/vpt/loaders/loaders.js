'use strict';
// src/js/loaders/{AbstractLoader,BlobLoader,LoaderFactory}.js.  BlobLoader wraps bytes in memory (Buffer, TypedArray or
// ArrayBuffer; the reference wraps a browser Blob); FileLoader is the local stand-in for AjaxLoader's ranged HTTP reads
// (AjaxLoader.js:20-26) — fs.read of [start, end) — networking itself is out of scope.
const fs = require('fs');

class AbstractLoader {
    async readLength() {}
    async readData(start, end) {}
}

class BlobLoader extends AbstractLoader {                       // BlobLoader.js:3-21

constructor(blob) {
    super();
    if (blob instanceof ArrayBuffer) { this.blob = new Uint8Array(blob); }
    else { this.blob = new Uint8Array(blob.buffer, blob.byteOffset, blob.byteLength); }
}

async readLength() { return this.blob.length; }

async readData(start, end) {                                    // Blob.slice clamps; .arrayBuffer() copies
    const b = this.blob.subarray(Math.max(start, 0), Math.max(end, 0));
    return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength);
}

}

class FileLoader extends AbstractLoader {

constructor(path) { super(); this.url = path; this._fd = fs.openSync(path, 'r'); this._length = fs.fstatSync(this._fd).size; }

close() { if (this._fd !== null) { fs.closeSync(this._fd); this._fd = null; } }

async readLength() { return this._length; }

async readData(start, end) {
    start = Math.min(Math.max(start, 0), this._length); end = Math.min(Math.max(end, start), this._length);
    const out = new Uint8Array(end - start);
    let done = 0;
    while (done < out.length) {
        const n = fs.readSync(this._fd, out, done, out.length - done, start + done);
        if (n <= 0) { throw new Error('short read from ' + this.url); }
        done += n;
    }
    return out.buffer;
}

}

function LoaderFactory(which) {                                 // LoaderFactory.js:4-12 ('ajax' is networking: not built)
    switch (which) {
        case 'blob': return BlobLoader;
        case 'file': return FileLoader;
        default: throw new Error('No suitable class');
    }
}

module.exports = { AbstractLoader, BlobLoader, FileLoader, LoaderFactory };
