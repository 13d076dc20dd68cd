'use strict';
// node js/bench.js [--volume 512] [--width 1920] [--height 1080] [--renderer mcm] [--tonemapper artistic] [--frames 200] [--fast-math 0]
// The Node host driving the path the way the reference's application does (RenderingContext.js:123-133,152-210):
// setVolume(reader) -> chooseRenderer -> chooseToneMapper -> N x render().  Prints one JSON line with ms/frame for the
// renderer alone and for renderer + tone mapper, and volume samples/s.  (The judged benchmark is bench.py; this shows the
// N-API host adds no per-frame cost of its own.)
const vpt = require('./vpt/index.js');
const { native } = require('./vpt/native.js');

function arg(name, dflt) {
    const i = process.argv.indexOf('--' + name);
    return i >= 0 ? process.argv[i + 1] : dflt;
}

function sphere(n) {                                            // radial falloff + a deterministic ripple, u8
    const v = new Uint8Array(n * n * n);
    for (let z = 0; z < n; z++) {
        const dz = (z + 0.5) / n - 0.5;
        for (let y = 0; y < n; y++) {
            const dy = (y + 0.5) / n - 0.5;
            for (let x = 0; x < n; x++) {
                const dx = (x + 0.5) / n - 0.5;
                const r = Math.sqrt(dx * dx + dy * dy + dz * dz);
                const base = 255 * Math.max(0, 1 - r / 0.45);
                const wob = r < 0.45 ? 40 * Math.sin(17 * dx) * Math.cos(13 * dy + 5 * dz) : 0;
                v[(z * n + y) * n + x] = Math.max(0, Math.min(255, Math.round(base + wob)));
            }
        }
    }
    return v;
}

function goldenRng() { let k = 1; return () => { const v = (k * 0.61803398875) % 1; k++; return v; }; }

async function main() {
    const n = Number(arg('volume', 512)), W = Number(arg('width', 1920)), H = Number(arg('height', 1080));
    const frames = Number(arg('frames', 200)), kind = arg('renderer', 'mcm'), tm = arg('tonemapper', 'artistic');
    const rc = new vpt.RenderingContext({ resolution: { width: W, height: H }, rng: goldenRng() });
    await rc.setVolume(new vpt.RAWReader(sphere(n), { width: n, height: n, depth: n }));
    rc.chooseRenderer(kind);
    rc.chooseToneMapper(tm);
    const N = native();
    // NO option call by default: what a maintainer who follows INTEGRATION.md gets from RendererFactory('mcm') + render() — the library's own
    // defaults (bit-exact contract arithmetic, tile classes, the HIT | MISS kernels on two streams).  --fast-math 1 adds the one opt-in of
    // bench.py's line (VPT_OPTION_FAST_MATH: hardware rcp / log / sin / cos, checked by tolerance)
    const fast = Number(arg('fast-math', 0));
    if (kind === 'mcm' && fast) { N.rendererSetOption(rc.renderer._h, N.VPT_OPTION_FAST_MATH, 1); }
    rc.renderer.reset();
    // warm up by TIME, like bench.py (a box that has just created its volume runs ~7 % slow for the first 0.2 s: clocks), then the median of three
    // synchronised blocks of `count` frames
    const time = (f, count) => {
        const w0 = process.hrtime.bigint();
        while (Number(process.hrtime.bigint() - w0) < 0.4e9) {
            for (let k = 0; k < 50; k++) { f(); }
            N.contextSynchronize(rc.gl._h);
        }
        const blocks = [];
        for (let b = 0; b < 3; b++) {
            const t0 = process.hrtime.bigint();
            for (let k = 0; k < count; k++) { f(); }
            const t1 = process.hrtime.bigint();
            N.contextSynchronize(rc.gl._h);
            const t2 = process.hrtime.bigint();
            blocks.push({ ms: Number(t2 - t0) / 1e6 / count, enqueue_ms: Number(t1 - t0) / 1e6 / count });
        }
        blocks.sort((x, y) => x.ms - y.ms);
        return blocks[1];
    };
    const a = time(() => rc.renderer.render(), frames);
    N.rendererClearSampleCount(rc.renderer._h);
    for (let k = 0; k < 10; k++) { rc.renderer.render(); }
    const samples = rc.renderer.sampleCount() / 10;
    const b = time(rc.render, frames);
    const frame = rc.getFrame();
    let lit = 0;
    for (let i = 0; i < W * H; i++) { if (frame.data[4 * i] > 0) { lit++; } }
    console.log(JSON.stringify({
        host: 'node ' + process.version, workload: kind + ' ' + n + '^3 @ ' + W + 'x' + H + ' + ' + tm, fast_math: kind === 'mcm' && !!fast, options_set: (kind === 'mcm' && fast) ? ['VPT_OPTION_FAST_MATH'] : [],
        renderer_ms_per_frame: a.ms, renderer_enqueue_ms_per_frame: a.enqueue_ms, samples_per_frame: samples,
        volume_samples_per_s: samples / (a.ms * 1e-3),
        renderer_plus_tonemapper_ms_per_frame: b.ms, lit_pixels: lit,
    }));
    rc.destroy();
}
main().catch(e => { console.error(e); process.exit(1); });
