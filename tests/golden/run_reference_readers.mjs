// Runs the reference's own readers (src/js/readers/{RAW,ZIP,BVP}Reader.js, imported in place through esm_loader.mjs)
// over the files named on the command line with a Buffer-backed loader, and prints what they return as JSON.
//   node --experimental-loader ./esm_loader.mjs run_reference_readers.mjs <archive.bvp> <volume.raw> <w> <h> <d>
import fs from 'fs';
import crypto from 'crypto';
import { RAWReader } from '/root/reference/src/js/readers/RAWReader.js';
import { ZIPReader } from '/root/reference/src/js/readers/ZIPReader.js';
import { BVPReader } from '/root/reference/src/js/readers/BVPReader.js';
import { ReaderFactory } from '/root/reference/src/js/readers/ReaderFactory.js';

class BufferLoader {                                           // the AbstractLoader interface (loaders/AbstractLoader.js:3-11)
    constructor(buf) { this.buf = buf; this.reads = []; }
    async readLength() { return this.buf.length; }
    async readData(start, end) {
        this.reads.push([start, end]);
        const b = this.buf.slice(start, end);
        return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength);
    }
}
const sha = ab => crypto.createHash('sha256').update(Buffer.from(ab)).digest('hex');

async function main() {
    const [, , archivePath, rawPath, w, h, d] = process.argv;
    const out = {};
    const archive = fs.readFileSync(archivePath);
    const zip = new ZIPReader(new BufferLoader(archive));
    out.zip_files = await zip.getFiles();
    out.zip_cd = zip._cd;
    out.zip_file_digests = {};
    for (const name of out.zip_files) {
        const data = await zip.readFile(name);
        out.zip_file_digests[name] = { length: data.byteLength, sha256: sha(data) };
    }
    try { await zip.readFile('missing.bin'); out.zip_missing = null; } catch (e) { out.zip_missing = e.message; }
    const bvpLoader = new BufferLoader(archive);
    const bvp = new BVPReader(bvpLoader);
    out.bvp_metadata = await bvp.readMetadata();
    out.bvp_blocks = [];
    for (let i = 0; i < out.bvp_metadata.blocks.length; i++) {
        const data = await bvp.readBlock(i);
        out.bvp_blocks.push({ length: data.byteLength, sha256: sha(data) });
    }
    const raw = fs.readFileSync(rawPath);
    const rr = new RAWReader(new BufferLoader(raw), { width: Number(w), height: Number(h), depth: Number(d) });
    out.raw_metadata = await rr.readMetadata();
    out.raw_blocks = [];
    for (let i = 0; i < Number(d); i++) {
        const data = await rr.readBlock(i);
        out.raw_blocks.push({ length: data.byteLength, sha256: sha(data) });
    }
    out.factory = { bvp: ReaderFactory('bvp') === BVPReader, raw: ReaderFactory('raw') === RAWReader, zip: ReaderFactory('zip') === ZIPReader };
    try { ReaderFactory('nrrd'); } catch (e) { out.factory_unknown = e.message; }
    process.stdout.write(JSON.stringify(out));
}
main().catch(e => { console.error(e); process.exit(1); });
