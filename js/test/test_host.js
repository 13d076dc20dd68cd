'use strict';
// node js/test/test_host.js — host-side checks that need no GPU: the addon loads, the matrix recipe matches the
// gl-matrix fixture bit for bit, PropertyBag / factory behave like the reference's.
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const vpt = require('../vpt/index.js');

const fixture = JSON.parse(fs.readFileSync(path.join(__dirname, '..', '..', 'tests', 'golden', 'mvp_inverse.json')));
for (const c of fixture.cases) {
    const cam = new vpt.Node();
    cam.transform.localTranslation = c.camera.translation;
    cam.transform.localRotation = c.camera.rotation;
    cam.transform.localScale = c.camera.scale;
    const pc = new vpt.PerspectiveCamera(cam, { fovy: c.fovy, aspect: c.aspect, near: c.near, far: c.far });
    cam.components.push(pc);
    const t = new vpt.Transform(new vpt.Node());
    t.localRotation = c.model.rotation; t.localTranslation = c.model.translation; t.localScale = c.model.scale;
    const m = vpt.mvpInverseMatrix(cam, t);
    const bits = Array.from(new Uint32Array(m.buffer));
    assert.deepStrictEqual(bits, c.inverse_bits, c.name);
    for (const key of ['iso_light', 'iso_light2']) {                                    // ISORenderer.js:152-166
        const l = vpt.isoLightDirection(cam, t, c[key]);
        assert.deepStrictEqual(Array.from(new Uint32Array(l.buffer)), c[key + '_bits'], c.name + ' ' + key);
    }
}
// CircleAnimator against the reference's own (tests/golden/circle_animator_r01.json)
const anim = JSON.parse(fs.readFileSync(path.join(__dirname, '..', '..', 'tests', 'golden', 'circle_animator_r01.json')));
for (const c of anim.cases) {
    const node = new vpt.Node();
    const a = new vpt.CircleAnimator(node, c.options);
    for (const f of c.frames) {
        a.update(f.t);
        assert.deepStrictEqual(Array.from(new Uint32Array(node.transform.localTranslation.buffer)), f.translation_bits);
        assert.deepStrictEqual(Array.from(new Uint32Array(node.transform.localRotation.buffer)), f.rotation_bits);
    }
}
// OrbitCameraAnimator against the reference's own, on scripted input (tests/golden/orbit_animator_r01.json)
{
    const orbit = JSON.parse(fs.readFileSync(path.join(__dirname, '..', '..', 'tests', 'golden', 'orbit_animator_r01.json')));
    let checked = 0;
    for (const c of orbit.cases) {
        const node = new vpt.Node();
        node.transform.localTranslation = c.start;
        let clock = 1000;
        const a = new vpt.OrbitCameraAnimator(node, null, Object.assign({}, c.options, { now: () => clock }));
        for (const f of c.frames) {
            const s = f.step, v0 = node.transform.version;
            let thrown = null;
            try {
                switch (s[0]) {
                    case 'rotate': a._rotateAroundFocus(s[1], s[2]); break;
                    case 'zoom': a._zoom(s[1]); break;
                    case 'move': a._move(s[1].slice()); break;
                    case 'pointerdown': a._handlePointerDown({ button: s[1] }); break;
                    case 'pointerup': a._handlePointerUp({}); break;
                    case 'pointermove': a._handlePointerMove({ movementX: s[1], movementY: s[2], shiftKey: s[3] }); break;
                    case 'wheel': a._handleWheel({ deltaY: s[1] }); break;
                    case 'keydown': a._handleKeyDown({ key: s[1] }); break;
                    case 'keyup': a._handleKeyUp({ key: s[1] }); break;
                    case 'tick': clock += s[1]; a._update(); break;
                }
            } catch (e) { thrown = e.constructor.name; }
            assert.strictEqual(thrown, f.throws);
            if (f.translation_bits === null) { assert.strictEqual(node.transform.version, v0); continue; }
            assert.deepStrictEqual(Array.from(new Uint32Array(node.transform.localTranslation.buffer)), f.translation_bits);
            assert.deepStrictEqual(Array.from(new Uint32Array(node.transform.localRotation.buffer)), f.rotation_bits);
            checked++;
        }
    }
    assert.ok(checked >= 15);
}
// PNG encoder: signature, IHDR, CRCs
{
    const png = vpt.encodePNG({ data: new Uint8Array([255, 0, 0, 255, 0, 255, 0, 255]), width: 1, height: 2 }, true);
    assert.strictEqual(png.slice(1, 4).toString('latin1'), 'PNG');
    assert.strictEqual(png.readUInt32BE(16), 1); assert.strictEqual(png.readUInt32BE(20), 2);
    assert.strictEqual(vpt.crc32(Buffer.from('IEND', 'latin1')), 0xae426082);
    const raw = require('zlib').inflateSync(png.slice(41, png.length - 16));        // the IDAT body
    assert.deepStrictEqual(Array.from(raw), [0, 0, 255, 0, 255, 0, 255, 0, 0, 255]);   // flipped: GL bottom row last
}
const bag = new vpt.PropertyBag();
bag.registerProperties([{ name: 'steps', value: 64 }]);
assert.strictEqual(bag.steps, 64);
let seen = null;
bag.addEventListener('change', e => { seen = e.detail; });
bag.dispatchEvent(new vpt.CustomEvent('change', { detail: { name: 'steps', value: 8 } }));
assert.deepStrictEqual(seen, { name: 'steps', value: 8 });
assert.throws(() => vpt.RendererFactory('nope'), /No suitable class/);
assert.strictEqual(vpt.RendererFactory('dos'), vpt.DOSRenderer);
assert.strictEqual(vpt.RendererFactory('lao'), vpt.LAORenderer);
assert.strictEqual(vpt.RendererFactory('iso'), vpt.ISORenderer); assert.strictEqual(vpt.RendererFactory('depth'), vpt.DepthRenderer);
assert.strictEqual(vpt.RendererFactory('mcm'), vpt.MCMRenderer);
// buffer spec hooks (AbstractRenderer.js:134-155): attachment counts and GL enums, without creating a native renderer
{
    const mk = (cls, size) => { const o = Object.create(cls.prototype); o._size = () => size; return o; };
    const m = mk(vpt.MCMRenderer, [40, 30]);
    assert.strictEqual(m._getAccumulationBufferSpec().length, 4);
    assert.strictEqual(m._getAccumulationBufferSpec()[3].iformat, 34836);          // RGBA32F
    assert.strictEqual(m._getFrameBufferSpec()[0].width, 40);
    assert.strictEqual(m._getRenderBufferSpec()[0].iformat, 34842);                // RGBA16F
    assert.strictEqual(mk(vpt.MIPRenderer, [8, 8])._getFrameBufferSpec()[0].iformat, 33321);   // R8
    assert.strictEqual(mk(vpt.DOSRenderer, [8, 8])._getAccumulationBufferSpec().length, 2);
    assert.strictEqual(mk(vpt.DepthRenderer, [8, 8])._getAccumulationBufferSpec()[0].iformat, 33326);   // R32F
}
// the transfer-function widget's data model (ui/TransferFunction/TransferFunction.js:74-85, 127-176) without a GPU: bump list, JSON round trip
{
    const tfx = JSON.parse(require('fs').readFileSync(require('path').join(__dirname, '../../tests/golden/tf_bumps_r04.json'), 'utf8'));
    const tf = new vpt.TransferFunction(null);
    assert.strictEqual(tf.transferFunctionWidth, 256); assert.strictEqual(tf.transferFunctionHeight, 256);
    assert.strictEqual(tf.addBump(), 0);
    assert.deepStrictEqual(tf.bumps, tfx.files.default_bump.bumps);
    tf.addBump({ position: { x: 0.1 }, color: { a: 0.25 } });
    assert.deepStrictEqual(tf.bumps[1], { position: { x: 0.1, y: 0.5 }, size: { x: 0.2, y: 0.2 }, color: { r: 1, g: 0, b: 0, a: 0.25 } });
    assert.deepStrictEqual(JSON.parse(tf.dumps()), tf.bumps);
    assert.deepStrictEqual(new vpt.TransferFunction(null).loads(tf.dumps()).bumps, tf.bumps);
    assert.strictEqual(tf.packed().length, 16);
    tf.removeBump(0); assert.strictEqual(tf.bumps.length, 1);
    tf.removeAllBumps(); assert.strictEqual(tf.packed().length, 0);
    assert.throws(() => new vpt.TransferFunction(null).loads('{"not":"an array"}'), TypeError);
    for (const e of Object.values(tfx.files)) assert.deepStrictEqual(new vpt.TransferFunction(null).loads(JSON.stringify(e.bumps)).bumps, e.bumps);
}
// the addon loads and reports the struct size the JS side packs
const { native } = require('../vpt/native.js');
assert.strictEqual(native().UNIFORMS_BYTES, vpt.U.SIZE);
assert.ok(/vpt/.test(native().version()));
console.log('js host ok:', fixture.cases.length, 'matrix cases');
