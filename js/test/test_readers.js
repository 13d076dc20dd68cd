'use strict';
// node js/test/test_readers.js — CPU: the Node host's loaders / readers against what the reference's own readers returned
// for the same bytes (tests/golden/readers_r01.json).  Run by tests/test_readers.py.
const assert = require('assert');
const crypto = require('crypto');
const fs = require('fs');
const os = require('os');
const path = require('path');
const vpt = require('../vpt/index.js');

const digest = ab => ({ length: ab.byteLength, sha256: crypto.createHash('sha256').update(Buffer.from(ab)).digest('hex') });

async function main() {
    const fx = JSON.parse(fs.readFileSync(path.join(__dirname, '../../tests/golden/readers_r01.json'), 'utf8'));
    const ref = fx.reference;
    const archive = Buffer.from(fx.archive_base64, 'base64'), raw = Buffer.from(fx.raw_base64, 'base64');
    const tmp = path.join(fs.mkdtempSync(path.join(os.tmpdir(), 'vpt-')), 'a.bvp');
    fs.writeFileSync(tmp, archive);
    for (const loader of [new vpt.BlobLoader(archive), new vpt.FileLoader(tmp)]) {
        const z = new vpt.ZIPReader(loader);
        assert.deepStrictEqual(await z.getFiles(), ref.zip_files);
        assert.deepStrictEqual(z._cd, ref.zip_cd);
        for (const name of ref.zip_files) { assert.deepStrictEqual(digest(await z.readFile(name)), ref.zip_file_digests[name]); }
        await assert.rejects(() => z.readFile('missing.bin'), e => e.message === ref.zip_missing);
    }
    fs.unlinkSync(tmp); fs.rmdirSync(path.dirname(tmp));
    const bvp = new vpt.BVPReader(new vpt.BlobLoader(archive));
    assert.deepStrictEqual(await bvp.readMetadata(), ref.bvp_metadata);
    for (let i = 0; i < ref.bvp_blocks.length; i++) { assert.deepStrictEqual(digest(await bvp.readBlock(i)), ref.bvp_blocks[i]); }
    const bvp2 = new vpt.BVPReader(new vpt.BlobLoader(archive));
    assert.deepStrictEqual(digest(await bvp2.readBlock(1)), ref.bvp_blocks[1]);            // BVPReader.js:23-25
    const [w, h, d] = fx.raw_dims_whd;
    const rr = new vpt.RAWReader(new vpt.BlobLoader(raw), { width: w, height: h, depth: d });
    assert.deepStrictEqual(await rr.readMetadata(), ref.raw_metadata);
    for (let i = 0; i < d; i++) { assert.deepStrictEqual(digest(await rr.readBlock(i)), ref.raw_blocks[i]); }
    assert.strictEqual(vpt.ReaderFactory('bvp'), vpt.BVPReader); assert.strictEqual(vpt.ReaderFactory('raw'), vpt.RAWReader);
    assert.strictEqual(vpt.ReaderFactory('zip'), vpt.ZIPReader);
    assert.throws(() => vpt.ReaderFactory('nrrd'), e => e.message === ref.factory_unknown);
    assert.strictEqual(vpt.LoaderFactory('blob'), vpt.BlobLoader);
    assert.throws(() => vpt.LoaderFactory('ftp'), /No suitable class/);
    console.log('js readers ok');
}
main().catch(e => { console.error(e); process.exit(1); });
