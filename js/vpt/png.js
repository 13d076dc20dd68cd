'use strict';
// Minimal PNG encoder (zlib only) for the animation recorder: 8-bit RGBA, filter type 0.
const zlib = require('zlib');

const CRC = (() => { const t = new Uint32Array(256); for (let n = 0; n < 256; n++) { let c = n; for (let k = 0; k < 8; k++) { c = (c & 1) ? (0xedb88320 ^ (c >>> 1)) : (c >>> 1); } t[n] = c >>> 0; } return t; })();
function crc32(buf) { let c = 0xffffffff; for (let i = 0; i < buf.length; i++) { c = CRC[(c ^ buf[i]) & 0xff] ^ (c >>> 8); } return (c ^ 0xffffffff) >>> 0; }
function chunk(tag, data) {
    const out = Buffer.alloc(12 + data.length);
    out.writeUInt32BE(data.length, 0); out.write(tag, 4, 'latin1'); data.copy(out, 8);
    out.writeUInt32BE(crc32(out.slice(4, 8 + data.length)), 8 + data.length);
    return out;
}
// image: { data: Uint8Array RGBA8 [height][width][4], width, height }; bottomUp: row 0 is the bottom row (GL convention)
function encodePNG(image, bottomUp) {
    const w = image.width, h = image.height, raw = Buffer.alloc(h * (1 + 4 * w));
    for (let j = 0; j < h; j++) {
        const src = (bottomUp === false ? j : h - 1 - j) * 4 * w;
        raw[j * (1 + 4 * w)] = 0;
        Buffer.from(image.data.buffer, image.data.byteOffset + src, 4 * w).copy(raw, j * (1 + 4 * w) + 1);
    }
    const ihdr = Buffer.alloc(13);
    ihdr.writeUInt32BE(w, 0); ihdr.writeUInt32BE(h, 4); ihdr[8] = 8; ihdr[9] = 6;
    return Buffer.concat([Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]), chunk('IHDR', ihdr),
                          chunk('IDAT', zlib.deflateSync(raw, { level: 6 })), chunk('IEND', Buffer.alloc(0))]);
}
module.exports = { encodePNG, crc32 };
