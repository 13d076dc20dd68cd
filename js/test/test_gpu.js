'use strict';
// node js/test/test_gpu.js — GPU: the Node.js host (js/vpt) through the N-API addon and the C-ABI, checked against the
// scalar JS ray-march (oracle/js/raymarch.js, test infrastructure).  Run by tests/test_js_gpu.py under -m gpu.
const assert = require('assert');
const vpt = require('../vpt/index.js');
const cpu = require('../../oracle/js/raymarch.js');
const { native } = require('../vpt/native.js');

function sphere(n) {
    const v = new Uint8Array(n * n * n);
    for (let z = 0; z < n; z++) { for (let y = 0; y < n; y++) { for (let x = 0; x < n; x++) {
        const dx = (x + 0.5) / n - 0.5, dy = (y + 0.5) / n - 0.5, dz = (z + 0.5) / n - 0.5;
        const r = Math.sqrt(dx * dx + dy * dy + dz * dz);
        const wob = 40 * Math.sin(17 * dx) * Math.cos(13 * dy + 5 * dz);
        v[(z * n + y) * n + x] = Math.max(0, Math.min(255, Math.round(255 * Math.max(0, 1 - r / 0.45) + (r < 0.45 ? wob : 0))));
    } } }
    return v;
}
function goldenRng() { let k = 1; return () => { const v = (k * 0.61803398875) % 1; k++; return v; }; }

async function main() {
    const N = native();
    const n = 32, W = 80, H = 48;
    const vol = sphere(n);
    const ctx = new vpt.Context(0);
    const volume = new vpt.Volume(ctx, new vpt.RAWReader(vol, { width: n, height: n, depth: n }));
    let progress = 0;
    volume.addEventListener('progress', e => { progress = e.detail; });
    await volume.load();                                           // RenderingContext.js:124-134
    volume.setFilter('linear');
    assert.strictEqual(progress, 1);
    const camera = vpt.defaultCamera(W / H);
    const transform = new vpt.Transform(new vpt.Node());
    const scene = new cpu.Scene(vol, n, n, n, 'linear', null, null);
    const mvpInv = vpt.mvpInverseMatrix(camera, transform);

    // ---- MIP: hooks one by one and fused, 3 frames
    for (const fused of [false, true]) {
        const R = vpt.RendererFactory('mip');
        const r = new R(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng(), fused });
        r.steps = 40;
        r.reset();
        const acc = new Uint8Array(W * H), rng = goldenRng();
        for (let k = 0; k < 3; k++) {
            r.render();
            cpu.mipRender(scene, { width: W, height: H, mvpInv, offset: rng(), steps: 40 }, acc);
        }
        const got = r.read(N.VPT_BUFFER_ACCUM, new Uint8Array(W * H));
        let bad = 0;
        for (let i = 0; i < W * H; i++) { if (got[i] !== acc[i]) { bad++; } }
        assert.ok(bad <= 1, 'MIP fused=' + fused + ': ' + bad + ' pixels differ');
        const tex = r.getTexture();
        assert.strictEqual(tex.width, W); assert.strictEqual(tex.height, H); assert.strictEqual(tex.data.length, W * H * 4);
        assert.ok(r.sampleCount() > 0);
        r.destroy();
    }

    // ---- MCM: reset + 4 passes, state buffers
    {
        const R = vpt.RendererFactory('mcm');
        const r = new R(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        r.extinction = 6; r.bounces = 3; r.steps = 5; r.anisotropy = 0.3;
        r.dispatchEvent(new vpt.CustomEvent('change', { detail: { name: 'extinction', value: 6 } }));    // Application.js:130-137 -> reset()
        const st = [0, 1, 2, 3].map(() => new Float32Array(W * H * 4)), rng = goldenRng();
        rng();                                                     // the constructor-time state is replaced by the 'change' reset
        const fr = { width: W, height: H, mvpInv, seed: 0, extinction: 6, anisotropy: 0.3, bounces: 3, steps: 5 };
        // two resets happened on the GPU side? no: one explicit reset() via the change event only
        const rng2 = goldenRng();
        fr.seed = rng2();
        cpu.mcmReset(fr, st);
        for (let k = 0; k < 4; k++) { r.render(); fr.seed = rng2(); cpu.mcmIntegrate(scene, fr, st); }
        const bufs = [N.VPT_BUFFER_MCM_POSITION, N.VPT_BUFFER_MCM_DIRECTION, N.VPT_BUFFER_MCM_TRANSMITTANCE, N.VPT_BUFFER_MCM_RADIANCE];
        let badPixels = new Set();
        bufs.forEach((b, bi) => {
            const got = new Uint32Array(r.read(b, new Float32Array(W * H * 4)).buffer), want = new Uint32Array(st[bi].buffer);
            for (let i = 0; i < got.length; i++) { if (got[i] !== want[i]) { badPixels.add(i >> 2); } }
        });
        assert.ok(badPixels.size <= 4, 'MCM: ' + badPixels.size + ' pixels differ');
        assert.strictEqual(r.sampleCount(), W * H * 5 * 4);
        r.destroy();
    }

    // ---- one-rank RCCL gather: sharded renderer + FrameGather == plain renderer, bit for bit
    {
        const opts = () => ({ resolution: { width: W, height: H }, transform, rng: goldenRng() });
        const plain = new vpt.MCMRenderer(ctx, volume, camera, null, opts());
        const sharded = new vpt.MCMRenderer(ctx, volume, camera, null, Object.assign(opts(), { shard: { rank: 0, world: 1, rows: 8 } }));
        plain.reset(); sharded.reset();
        const gather = new vpt.FrameGather(sharded, vpt.FrameGather.uniqueId(), 0, 1, 0);
        for (let k = 0; k < 3; k++) { plain.render(); gather.render(); }
        gather.synchronize();
        const want = plain.getTexture(), got = gather.getFrame();
        assert.strictEqual(got.width, W); assert.strictEqual(got.height, H);
        assert.deepStrictEqual(got.data, want.data);
        gather.destroy(); plain.destroy(); sharded.destroy();
    }

    // ---- tone mappers: the committed contract fixture (tests/golden/tonemap_r01.json), then renderer -> tone mapper in HBM
    {
        const fx = JSON.parse(require('fs').readFileSync(require('path').join(__dirname, '../../tests/golden/tonemap_r01.json'), 'utf8'));
        const src = Uint16Array.from(fx.source_rgba16f_bits);
        const n = src.length / 4;
        for (const c of fx.cases) {
            const T = vpt.ToneMapperFactory(c.kind);
            const tm = new T(ctx, { data: src, width: n, height: 1 }, { resolution: { width: n, height: 1 } });
            Object.keys(c.params).forEach(k => { tm[k] = c.params[k]; });
            tm.render();
            const out = tm.getTexture();
            assert.strictEqual(out.data.length, c.rgba8.length);
            assert.deepStrictEqual(Array.from(out.data), c.rgba8, 'tone mapper ' + c.kind + ' ' + JSON.stringify(c.params));
            tm.destroy();
        }
        assert.throws(() => vpt.ToneMapperFactory('linear'), /No suitable class/);           // ToneMapperFactory.js:26
        const r = new vpt.MCMRenderer(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        r.reset();
        const tm = new (vpt.ToneMapperFactory('artistic'))(ctx, r, { resolution: { width: W, height: H } });   // RenderingContext.js:169-188
        assert.deepStrictEqual(tm.properties.map(p => p.name), ['low', 'high', 'mid', 'saturation', 'gamma']);
        for (let k = 0; k < 3; k++) { r.render(); tm.render(); }                                               // RenderingContext.js:196-197
        const out = tm.getTexture();
        assert.strictEqual(out.width, W); assert.strictEqual(out.height, H);
        let lit = 0;
        for (let i = 0; i < W * H; i++) { assert.strictEqual(out.data[4 * i + 3], 255); if (out.data[4 * i] > 0) { lit++; } }
        assert.ok(lit > W * H / 2);
        tm.destroy(); r.destroy();
    }

    // ---- the caller's sequence through the headless RenderingContext (RenderingContext.js:123-133,152-210)
    {
        const rc = new vpt.RenderingContext({ resolution: { width: W, height: H }, rng: goldenRng() });
        let last = 0;
        rc.addEventListener('progress', e => { last = e.detail; });
        await rc.setVolume(new vpt.RAWReader(vol, { width: n, height: n, depth: n }));
        assert.strictEqual(last, 1);
        rc.render();                                                   // nothing chosen yet: a no-op
        rc.chooseRenderer('mcm'); rc.chooseToneMapper('reinhard');
        for (let k = 0; k < 3; k++) { rc.render(); }
        const f = rc.getFrame();
        assert.strictEqual(f.width, W); assert.strictEqual(f.height, H);
        let lit = 0;
        for (let i = 0; i < W * H; i++) { assert.strictEqual(f.data[4 * i + 3], 255); if (f.data[4 * i] > 0) { lit++; } }
        assert.ok(lit > W * H / 2);
        rc.resolution = { width: 40, height: 32 }; rc.resize(40, 32);
        rc.chooseRenderer('mip'); rc.setFilter('nearest');
        rc.render();
        assert.strictEqual(rc.getFrame().data.length, 40 * 32 * 4);
        rc.destroy();
    }

    // ---- a two-channel (RG8) volume through the Node host: with G = 0 it must render exactly like the R8 volume
    {
        const rg = new Uint8Array(2 * n * n * n);
        for (let i = 0; i < n * n * n; i++) { rg[2 * i] = vol[i]; }
        const reader = {
            async readMetadata() {
                const md = await new vpt.RAWReader(vol, { width: n, height: n, depth: n }).readMetadata();
                md.modalities[0].format = vpt.GL_RG; md.modalities[0].internalFormat = vpt.GL_RG8;
                return md;
            },
            async readBlock(i) { return rg.subarray(2 * i * n * n, 2 * (i + 1) * n * n); },
        };
        const v2 = new vpt.Volume(ctx, reader);
        await v2.load(); v2.setFilter('linear');
        const imgs = [volume, v2].map(v => {
            const r = new vpt.EAMRenderer(ctx, v, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
            r.reset(); r.render(); r.render();
            const out = r.getTexture().data;
            r.destroy();
            return out;
        });
        assert.deepStrictEqual(imgs[1], imgs[0], 'RG8 with G = 0 equals R8');
        v2.destroy();
    }

    // ---- frame sequences: play(count, FUSED) == count x render(), bit for bit (MCM state, MCS accumulator)
    for (const kind of ['mcm', 'mcs']) {
        const mk = () => { const R = vpt.RendererFactory(kind); const r = new R(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() }); r.extinction = 6; r.reset(); return r; };
        const a = mk(), b = mk();
        for (let k = 0; k < 6; k++) { a.render(); }
        b.play(4, N.VPT_PLAY_FUSED); b.play(2, N.VPT_PLAY_EAGER);
        assert.deepStrictEqual(b.getTexture().data, a.getTexture().data, kind + ' play');
        assert.strictEqual(b.sampleCount(), a.sampleCount());
        a.destroy(); b.destroy();
    }

    // ---- play(count, FRAMES) (MCM): slot f of the frame ring == the render buffer after the f-th of count render() calls
    {
        const mk = () => { const r = new vpt.MCMRenderer(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() }); r.extinction = 6; r.reset(); return r; };
        const a = mk(), b = mk();
        b.play(5, N.VPT_PLAY_FRAMES);
        for (let f = 0; f < 5; f++) {
            a.render();
            assert.deepStrictEqual(b.readFrameSlot(f).data, a.getTexture().data, 'frame slot ' + f);
        }
        assert.deepStrictEqual(b.getTexture().data, a.getTexture().data, 'render buffer after a frame sequence');
        assert.strictEqual(b.sampleCount(), a.sampleCount());
        assert.throws(() => b.readFrameSlot(5));
        assert.throws(() => b.play(N.VPT_FRAME_SLOTS + 1, N.VPT_PLAY_FRAMES));
        a.destroy(); b.destroy();
    }

    // ---- ISO and Depth through the Node host (uniform block offsets 112..124): hits, misses and shading present
    {
        const iso = new vpt.ISORenderer(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        assert.deepStrictEqual(iso.properties.map(p => p.name), ['steps', 'isovalue', 'light', 'transferFunction']);
        iso.isovalue = 0.3;
        iso.reset(); iso.render(); iso.render();
        const closest = new Uint16Array(W * H * 4);
        iso.read(N.VPT_BUFFER_ACCUM, closest);
        let hits = 0, misses = 0;
        for (let i = 0; i < W * H; i++) { if (closest[4 * i + 3] === 0xbc00) { misses++; } else if (closest[4 * i + 3] < 0x3c01) { hits++; } }
        assert.ok(hits > 50 && misses > 0 && hits + misses === W * H, 'iso hits ' + hits + ' misses ' + misses);
        const img = iso.getTexture().data;
        let shaded = 0;
        for (let i = 0; i < W * H; i++) { if (img[4 * i] !== 0x3c00 && img[4 * i] !== 0) { shaded++; } }
        assert.ok(shaded > 0);
        assert.ok(iso.sampleCount() >= 2 * hits * 50);
        iso.destroy();
        const dep = new vpt.DepthRenderer(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        assert.deepStrictEqual(dep.properties.map(p => p.name), ['extinction', 'slices', 'threshold', 'random', 'transferFunction']);
        dep.reset(); dep.render(); dep.render();
        const d = new Float32Array(W * H);
        dep.read(N.VPT_BUFFER_ACCUM, d);
        let pos = 0, neg = 0;
        for (let i = 0; i < W * H; i++) { if (d[i] > 0) { pos++; } else if (d[i] < 0) { neg++; } }
        assert.ok(pos > 50 && neg > 0, 'depth pos ' + pos + ' neg ' + neg);
        dep.destroy();
        // LAO: the parameter block reaches the kernels (occlusion and shadows only ever darken), alpha is 1, frames replace
        const lao = new vpt.LAORenderer(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        assert.strictEqual(lao.properties.length, 13);
        lao.slices = 24;
        const shot = () => { lao.reset(); lao.render(); const a = new Uint8Array(W * H * 4); lao.read(N.VPT_BUFFER_ACCUM, a); return a; };
        const lit = shot();
        lao.localAmbientOcclusion = false; lao.softShadows = false;
        const plain = shot();
        let darker = 0, brighter = 0, lum = 0;
        for (let i = 0; i < W * H; i++) {
            assert.strictEqual(lit[4 * i + 3], 255);
            for (let c = 0; c < 3; c++) { lum += plain[4 * i + c]; if (lit[4 * i + c] < plain[4 * i + c]) { darker++; } else if (lit[4 * i + c] > plain[4 * i + c]) { brighter++; } }
        }
        assert.ok(lum > 0 && darker > 100 && darker > 20 * brighter, 'lao darker ' + darker + ' brighter ' + brighter);
        lao.render();
        const again = new Uint8Array(W * H * 4); lao.read(N.VPT_BUFFER_ACCUM, again);
        assert.deepStrictEqual(again, plain);                  // no accumulation across frames (LAORenderer.glsl:225-227)
        lao.LAOStepSize = 0;
        assert.throws(() => lao.render(), /step size/);
        lao.destroy();
        // DOS: the slice sweep driven by the Node host: progressive, ends past the far corner, occlusion darkens, repeatable
        const mk = () => new vpt.DOSRenderer(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        const dos = mk();
        assert.deepStrictEqual(dos.properties.map(p => p.name), ['steps', 'slices', 'extinction', 'aperture', 'samples', 'transferFunction']);
        assert.strictEqual(dos._occlusionSamples.length, 16);
        dos.slices = 40; dos.steps = 25;
        dos.reset();
        assert.ok(dos._minDepth > 0 && dos._maxDepth > dos._minDepth);
        let nslices = 0;
        for (let k = 0; k < 3; k++) { dos.render(); nslices += dos._slices.length / 3; }
        assert.ok((nslices === 40 || nslices === 41) && dos._slices.length === 0, 'dos slices ' + nslices);
        const col = new Float32Array(W * H * 4), occ = new Float32Array(W * H);
        dos.read(N.VPT_BUFFER_ACCUM, col); dos.read(N.VPT_BUFFER_DOS_OCCLUSION, occ);
        let opaque = 0, empty = 0, shadowed = 0;
        for (let i = 0; i < W * H; i++) {
            assert.ok(occ[i] >= 0 && occ[i] <= 1 && col[4 * i + 3] >= 0 && col[4 * i + 3] <= 1);
            if (col[4 * i + 3] > 0.5) { opaque++; } else if (col[4 * i + 3] === 0) { empty++; }
            if (occ[i] < 0.5) { shadowed++; }
        }
        assert.ok(opaque > 50 && empty > 50 && shadowed > 50, 'dos opaque ' + opaque + ' empty ' + empty + ' shadowed ' + shadowed);
        const dosImg = dos.getTexture().data;
        const dos2 = mk(); dos2.slices = 40; dos2.steps = 25; dos2.reset();
        for (let k = 0; k < 3; k++) { dos2.render(); }
        assert.deepStrictEqual(dos2.getTexture().data, dosImg);
        assert.throws(() => dos.play(2), /frame sequences/);
        assert.throws(() => N.rendererIntegrate(dos._h, dos._u), /integrate_slices/);
        dos.destroy(); dos2.destroy();
    }

    // ---- errors are thrown Errors carrying the native message
    {
        const r = new vpt.MIPRenderer(ctx, null, camera, null, { resolution: 32 });
        r.reset();
        assert.throws(() => r.render(), /no ready volume/);
        r.destroy();
    }
    // ---- manifests beyond R8 (Volume.js:58-60,84-105): an R32F and an R16F volume with the same texels, and an RGBA8 volume against
    //      its first two channels as RG8, must render identically; integer types throw the reference's error
    {
        const GL = require('../vpt/readers/readers.js');
        const f32 = new Float32Array(n * n * n), f16 = new Uint16Array(n * n * n), rgba = new Uint8Array(4 * n * n * n), rg = new Uint8Array(2 * n * n * n);
        const toHalf = v => { const f = new Float32Array([v]), u = new Uint32Array(f.buffer)[0]; const e = ((u >> 23) & 255) - 112, m = (u >> 13) & 1023;
            return e <= 0 ? 0 : ((u >> 16) & 0x8000) | (e << 10) | m; };     // truncating: only used to MAKE a half file
        const halfVal = h => { const e = (h >> 10) & 31, m = h & 1023; return e === 0 ? m * Math.pow(2, -24) : (1 + m / 1024) * Math.pow(2, e - 15); };
        for (let i = 0; i < vol.length; i++) {
            f16[i] = toHalf(vol[i] / 255 * 1.3); f32[i] = halfVal(f16[i]);
            rgba[4 * i] = vol[i]; rgba[4 * i + 1] = 255 - vol[i]; rgba[4 * i + 2] = 7; rgba[4 * i + 3] = 200;
            rg[2 * i] = vol[i]; rg[2 * i + 1] = 255 - vol[i];
        }
        const mk = (bytes, format, internalFormat, type) => {
            const rd = new vpt.RAWReader(new Uint8Array(bytes.buffer), { width: n, height: n, depth: n });
            const base = rd.readMetadata.bind(rd), bpv = bytes.byteLength / (n * n * n);
            rd.readMetadata = async () => { const m = await base(); m.modalities[0].format = format; m.modalities[0].internalFormat = internalFormat; m.modalities[0].type = type; return m; };
            rd.readBlock = i => new Uint8Array(bytes.buffer, i * n * n * bpv, n * n * bpv);
            return new vpt.Volume(ctx, rd);
        };
        const render = async v => {
            await v.load(); v.setFilter('linear');
            const r = new (vpt.RendererFactory('eam'))(ctx, v, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
            r.reset(); r.render(); r.render();
            const out = r.read(N.VPT_BUFFER_ACCUM, new Uint8Array(4 * W * H));
            r.destroy(); v.destroy();
            return out;
        };
        const a = await render(mk(f32, GL.GL_RED, GL.GL_R32F, GL.GL_FLOAT));
        const b = await render(mk(f16, GL.GL_RED, GL.GL_R16F, GL.GL_HALF_FLOAT));
        assert.deepStrictEqual(Array.from(b), Array.from(a), 'R16F and R32F volumes with the same texels');
        assert.ok(a.some(x => x > 0 && x < 255));
        const c = await render(mk(rgba, GL.GL_RGBA, GL.GL_RGBA8, GL.GL_UNSIGNED_BYTE));
        const d = await render(mk(rg, GL.GL_RG, GL.GL_RG8, GL.GL_UNSIGNED_BYTE));
        assert.deepStrictEqual(Array.from(c), Array.from(d), 'RGBA8 keeps the two channels the shaders read');
        let threw = false;
        try { await mk(new Uint16Array(n * n * n), GL.GL_RED, 33322, 5123).load(); } catch (e) { threw = /Unknown volume datatype/.test(e.message); }
        assert.ok(threw, 'UNSIGNED_SHORT volumes raise the reference error');
    }
    {   // the transfer-function widget as data: the bump files of tests/golden/tf_bumps_r04.json -> the texels the oracle's restatement makes
        const crypto = require('crypto');
        const fx = JSON.parse(require('fs').readFileSync(require('path').join(__dirname, '../../tests/golden/tf_bumps_r04.json'), 'utf8'));
        for (const [name, entry] of Object.entries(fx.files)) {
            for (const [key, want] of Object.entries(entry.sha256)) {
                const [size, form] = key.split('_'), [w, h] = size.split('x').map(Number);
                const tf = new vpt.TransferFunction(ctx, entry.bumps, w, h);
                const t = tf.texture(form === 'unpremultiplied');
                assert.strictEqual(t.width, w); assert.strictEqual(t.height, h); assert.strictEqual(t.data.length, w * h * 4);
                assert.strictEqual(crypto.createHash('sha256').update(t.data).digest('hex'), want, 'transfer function ' + name + ' ' + key);
            }
        }
        const tf = new vpt.TransferFunction(ctx);
        assert.strictEqual(tf.addBump(), 0);
        assert.deepStrictEqual(tf.bumps, fx.files.default_bump.bumps);                 // addBump() defaults (TransferFunction.js:129-144)
        assert.deepStrictEqual(new vpt.TransferFunction(ctx).loads(tf.dumps()).bumps, tf.bumps);
        const r = new (vpt.RendererFactory('mip'))(ctx, volume, camera, null, { resolution: { width: W, height: H }, transform, rng: goldenRng() });
        r.setTransferFunction(tf.value);                                               // what Application.js does with the widget's canvas
        r.reset(); r.render();
        r.destroy();
        let threw = false;
        try { new vpt.TransferFunction(ctx, [{ size: { x: 0 } }]).texture(); } catch (e) { threw = /zero size/.test(e.message); }
        assert.ok(threw, 'a bump of zero size is refused');
    }
    volume.destroy();
    ctx.destroy();
    console.log('js gpu ok');
}
main().catch(e => { console.error(e); process.exit(1); });
