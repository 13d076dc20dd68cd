'use strict';
/*
 * raymarch.js — scalar JavaScript ray-march of the MIP / EAM / MCS / MCM passes (CPU).
 *
 * TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT: used by tests (config C1 "plumbing" run) and by bench.py's
 * cpu_baseline leg (the scalar JS baseline BASELINE.json asks for).  The product path (js/vpt, vpt_amd) never
 * requires this file.
 *
 * It follows the same numeric contract as oracle/vpt_oracle.c (DESIGN.md §3), emulating binary32 with Math.fround
 * after every operation.  fmaf(a,b,c) is emulated as fround(a*b + c) in binary64: the product is exact, the sum is
 * rounded twice, so a result can differ from a true fma in the last bit about once in 2^29 operations — outputs
 * agree with the C oracle bit for bit except for isolated pixels (tests allow 1e-4 of them).
 * Reference files followed: src/glsl/renderers/{MIP,EAM,MCS,MCM}Renderer.glsl and the mixins they include
 * (cited per function in oracle/vpt_oracle.c, whose structure this mirrors).
 */
const f = Math.fround;
const fma = (a, b, c) => f(a * b + c);
const cvt = new DataView(new ArrayBuffer(4));
const f2u = x => { cvt.setFloat32(0, x); return cvt.getUint32(0); };
const u2f = x => { cvt.setUint32(0, x >>> 0); return cvt.getFloat32(0); };

// ---- GLSL built-ins (minNum / maxNum contract) ----------------------------------------------------------------
function vmin(a, b) {
    if (a !== a) { return b; }
    if (b !== b) { return a; }
    if (a === 0 && b === 0) { return Object.is(a, -0) ? a : b; }
    return b < a ? b : a;
}
function vmax(a, b) {
    if (a !== a) { return b; }
    if (b !== b) { return a; }
    if (a === 0 && b === 0) { return Object.is(a, -0) ? b : a; }
    return a < b ? b : a;
}
const clamp01 = x => vmin(vmax(x, 0), 1);
const mixf = (a, b, t) => fma(b, t, f(a * f(1 - t)));
const lerpf = (a, b, t) => fma(t, f(b - a), a);
const dot3 = (ax, ay, az, bx, by, bz) => fma(az, bz, fma(ay, by, f(ax * bx)));

function rcpNr(x) {
    let r = u2f((0x7EF311C7 - f2u(x)) >>> 0);
    r = fma(fma(-x, r, 1), r, r); r = fma(fma(-x, r, 1), r, r); r = fma(fma(-x, r, 1), r, r);
    return r;
}
function rcpNrz(x) {
    let r = u2f((0x7EF311C7 - f2u(x)) >>> 0);
    r = fma(fma(-x, r, 1), r, r); r = fma(fma(-x, r, 1), r, r);
    const M = 3.4028234663852886e+38;
    if (r !== r) { r = -M; } if (r > M) { r = M; } if (r < -M) { r = -M; }
    r = fma(fma(-x, r, 1), r, r);
    return r;
}
function rsqrtNr(x) {
    let y = u2f((0x5F375A86 - (f2u(x) >>> 1)) >>> 0);
    const h = f(0.5 * x);
    y = f(y * fma(f(-h * y), y, 1.5)); y = f(y * fma(f(-h * y), y, 1.5)); y = f(y * fma(f(-h * y), y, 1.5));
    return y;
}
const sqrtNr = x => f(x * rsqrtNr(x));
function logf(x) {
    if (x === 0) { return -Infinity; }
    if (!(x > 0)) { return NaN; }
    if (x === Infinity) { return Infinity; }
    const b = f2u(x);
    let e = (b >>> 23) - 126;
    let m = u2f((b & 0x007fffff) | 0x3f000000);
    if (m < f(0.70710678118654752440)) { e -= 1; m = f(f(m + m) - 1); } else { m = f(m - 1); }
    const fe = e, z = f(m * m);
    let p = f(7.0376836292E-2);
    p = fma(p, m, f(-1.1514610310E-1)); p = fma(p, m, f(1.1676998740E-1)); p = fma(p, m, f(-1.2420140846E-1));
    p = fma(p, m, f(1.4249322787E-1)); p = fma(p, m, f(-1.6668057665E-1)); p = fma(p, m, f(2.0000714765E-1));
    p = fma(p, m, f(-2.4999993993E-1)); p = fma(p, m, f(3.3333331174E-1));
    let y = f(f(p * m) * z);
    y = fma(f(-2.12194440e-4), fe, y);
    y = fma(-0.5, z, y);
    let r = f(m + y);
    r = fma(0.693359375, fe, r);
    return r;
}
const SC = { s: 0, c: 0 };
function sincosf(a) {
    const q = Math.round(f(a * f(0.63661977236758134308)));       // inputs are >= 0 and never a .5 tie in practice
    let r = fma(q, -1.5703125, a);
    r = fma(q, f(-4.837512969970703125e-4), r);
    r = fma(q, f(-7.54978995489188216e-8), r);
    const r2 = f(r * r);
    let ps = f(-1.9515295891E-4); ps = fma(ps, r2, f(8.3321608736E-3)); ps = fma(ps, r2, f(-1.6666654611E-1));
    const sr = fma(f(ps * r2), r, r);
    let pc = f(2.443315711809948E-005); pc = fma(pc, r2, f(-1.388731625493765E-003)); pc = fma(pc, r2, f(4.166664568298827E-002));
    const cr = fma(f(pc * r2), r2, fma(-0.5, r2, 1));
    const n = q & 3;
    if (n === 0) { SC.s = sr; SC.c = cr; } else if (n === 1) { SC.s = cr; SC.c = -sr; }
    else if (n === 2) { SC.s = -sr; SC.c = -cr; } else { SC.s = -cr; SC.c = sr; }
}

// ---- RNG (pcg.glsl:3-7, squashlinear.glsl:7-9, uniformdivision.glsl:3-6) ---------------------------------------
function pcg(x) {
    x = (Math.imul(x, 747796405) + 2891336453) >>> 0;
    x = Math.imul(((x >>> ((x >>> 28) + 4)) ^ x) >>> 0, 277803737) >>> 0;
    return ((x >>> 22) ^ x) >>> 0;
}
const hash3 = (x, y, z) => pcg((Math.imul(19, x) + Math.imul(47, y) + Math.imul(101, z) + 131) >>> 0);
const RNG = { state: 0 };
function uniform() { RNG.state = pcg(RNG.state); return f(f(RNG.state) * 2.3283064365386963e-10); }
const exponential = invRate => f(-logf(uniform()) * invRate);

// ---- scene ---------------------------------------------------------------------------------------------------
const SRGB = new Float32Array(256);
for (let c = 0; c < 256; c++) { const s = c / 255; SRGB[c] = s <= 0.04045 ? s / 12.92 : Math.pow((s + 0.055) / 1.055, 2.4); }
const INV255 = f(0.00392156862745098);

class Scene {
    // volume: Uint8Array [z][y][x]; tf / env: { data: Uint8Array RGBA8, width, height } (defaults of the reference if null)
    constructor(volume, nx, ny, nz, filter, tf, env) {
        this.vol = volume; this.nx = nx; this.ny = ny; this.nz = nz; this.linear = filter !== 'nearest';
        tf = tf || { data: new Uint8Array([255, 0, 0, 0, 255, 0, 0, 255]), width: 2, height: 1 };     // AbstractRenderer.js:34
        env = env || { data: new Uint8Array([255, 255, 255, 255]), width: 1, height: 1 };             // RenderingContext.js:93
        this.tfW = tf.width;
        this.tf = new Float32Array(tf.width * 4);                // row 0 (R8 volume: lookup at (r, 0))
        for (let i = 0; i < tf.width; i++) {
            for (let c = 0; c < 3; c++) { this.tf[4 * i + c] = SRGB[tf.data[4 * i + c]]; }
            this.tf[4 * i + 3] = f(tf.data[4 * i + 3] / 255);
        }
        if (env.width !== 1 || env.height !== 1) { throw new Error('raymarch.js supports the 1x1 environment only'); }
        this.env = [f(env.data[0] / 255), f(env.data[1] / 255), f(env.data[2] / 255), f(env.data[3] / 255)];
        this.samples = 0;
    }
}
const COL = new Float32Array(4);
function linearCoord(s, n) {          // returns i0 (clamped); sets LC.i1, LC.f
    let u = fma(s, n, -0.5);
    if (!(u > -1)) { u = -1; }
    if (u > n) { u = n; }
    const fl = Math.floor(u);
    LC.f = f(u - fl);
    const i = fl;
    LC.i1 = i + 1 < 0 ? 0 : (i + 1 > n - 1 ? n - 1 : i + 1);
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
const LC = { i1: 0, f: 0 };
function sampleVolumeColor(sc, px, py, pz) {
    const v = sc.vol, nx = sc.nx, ny = sc.ny, nz = sc.nz, sy = nx, sz = nx * ny;
    let r;
    sc.samples++;
    if (!sc.linear) {
        const near = (s, n) => { let u = f(s * n); if (!(u > 0)) { u = 0; } if (u > n - 1) { u = n - 1; } return Math.floor(u); };
        r = f(v[near(px, nx) + near(py, ny) * sy + near(pz, nz) * sz] * INV255);
    } else {
        const x0 = linearCoord(px, nx), x1 = LC.i1, fx = LC.f;
        const y0 = linearCoord(py, ny), y1 = LC.i1, fy = LC.f;
        const z0 = linearCoord(pz, nz), z1 = LC.i1, fz = LC.f;
        const c00 = lerpf(v[x0 + y0 * sy + z0 * sz], v[x1 + y0 * sy + z0 * sz], fx), c10 = lerpf(v[x0 + y1 * sy + z0 * sz], v[x1 + y1 * sy + z0 * sz], fx);
        const c01 = lerpf(v[x0 + y0 * sy + z1 * sz], v[x1 + y0 * sy + z1 * sz], fx), c11 = lerpf(v[x0 + y1 * sy + z1 * sz], v[x1 + y1 * sy + z1 * sz], fx);
        r = f(lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz) * INV255);
    }
    const i0 = linearCoord(r, sc.tfW), i1 = LC.i1, ft = LC.f, t = sc.tf;
    COL[0] = lerpf(t[4 * i0], t[4 * i1], ft); COL[1] = lerpf(t[4 * i0 + 1], t[4 * i1 + 1], ft);
    COL[2] = lerpf(t[4 * i0 + 2], t[4 * i1 + 2], ft); COL[3] = lerpf(t[4 * i0 + 3], t[4 * i1 + 3], ft);
}

// ---- ray set-up ------------------------------------------------------------------------------------------------
const pixelNdc = (i, n) => f(f((2 * i + 1) / n) - 1);
const ndcToUv = p => fma(p, 0.5, 0.5);
const P = new Float32Array(3), Q = new Float32Array(3);
function unprojectPoint(m, x, y, z, out) {
    const rx = fma(m[4], y, fma(m[0], x, fma(m[8], z, m[12]))), ry = fma(m[5], y, fma(m[1], x, fma(m[9], z, m[13])));
    const rz = fma(m[6], y, fma(m[2], x, fma(m[10], z, m[14]))), rw = fma(m[7], y, fma(m[3], x, fma(m[11], z, m[15])));
    const i = rcpNr(rw);
    out[0] = f(rx * i); out[1] = f(ry * i); out[2] = f(rz * i);
}
const TB = { near: 0, far: 0 };
function intersectCube(ox, oy, oz, dx, dy, dz) {
    const ix = rcpNrz(dx), iy = rcpNrz(dy), iz = rcpNrz(dz);
    const ax = f(f(0 - ox) * ix), bx = f(f(1 - ox) * ix), ay = f(f(0 - oy) * iy), by = f(f(1 - oy) * iy);
    const az = f(f(0 - oz) * iz), bz = f(f(1 - oz) * iz);
    TB.near = vmax(vmax(vmin(ax, bx), vmin(ay, by)), vmin(az, bz));
    TB.far = vmin(vmin(vmax(ax, bx), vmax(ay, by)), vmax(az, bz));
}
const toUnorm8 = x => { const c = clamp01(x); return rintEven(f(c * 255)); };
function rintEven(x) { const r = Math.round(x); return (r - x === 0.5 && (r & 1)) ? r - 1 : r; }

// ---- MIP (MIPRenderer.glsl:51-72, :105-109) -----------------------------------------------------------------------
// frame params: { width, height, mvpInv: Float32Array(16), offset, steps }
function mipRender(sc, fr, acc) {                       // acc: Uint8Array(width*height), max-accumulated in place
    const W = fr.width, H = fr.height, m = fr.mvpInv, step = f(1 / fr.steps), off0 = f(fr.offset);
    for (let j = 0; j < H; j++) {
        for (let i = 0; i < W; i++) {
            const px = pixelNdc(i, W), py = pixelNdc(j, H);
            unprojectPoint(m, px, py, -1, P); unprojectPoint(m, px, py, 1, Q);
            intersectCube(P[0], P[1], P[2], f(Q[0] - P[0]), f(Q[1] - P[1]), f(Q[2] - P[2]));
            const t0 = vmax(TB.near, 0), t1 = vmax(TB.far, 0);
            let out = 0;
            if (!(t0 >= t1)) {
                const fx = mixf(P[0], Q[0], t0), fy = mixf(P[1], Q[1], t0), fz = mixf(P[2], Q[2], t0);
                const tx = mixf(P[0], Q[0], t1), ty = mixf(P[1], Q[1], t1), tz = mixf(P[2], Q[2], t1);
                let t = 0, val = 0, offset = off0;
                do {
                    sampleVolumeColor(sc, mixf(fx, tx, offset), mixf(fy, ty, offset), mixf(fz, tz, offset));
                    val = vmax(COL[3], val);
                    t = f(t + step);
                    const mm = f(offset + step);
                    offset = f(mm - Math.floor(mm));
                } while (t < 1);
                out = val;
            }
            const q = toUnorm8(out), k = j * W + i;
            if (q > acc[k]) { acc[k] = q; }
        }
    }
}

// ---- MCM (MCMRenderer.glsl:70-78, :116-172, :259-275; unprojectRand.glsl:3-24) -----------------------------------
// state: 4 Float32Array(width*height*4): [pos,0] [dir,bounces] [transmittance,0] [radiance,samples]
const PH = { px: 0, py: 0, pz: 0, dx: 0, dy: 0, dz: 0, tx: 1, ty: 1, tz: 1, bounces: 0 };
function resetPhoton(fr, ndcx, ndcy) {
    const m = fr.mvpInv;
    uniform(); uniform();                               // random_disk(state) * blur with blur == 0
    unprojectPoint(m, f(ndcx + 0), f(ndcy + 0), -1, P);
    const sx = uniform(), sy = uniform();
    const ax = f(fma(sx, 2, -1) * fr.invW), ay = f(fma(sy, 2, -1) * fr.invH);
    unprojectPoint(m, f(ndcx + ax), f(ndcy + ay), 1, Q);
    let dx = f(Q[0] - P[0]), dy = f(Q[1] - P[1]), dz = f(Q[2] - P[2]);
    const inv = rsqrtNr(dot3(dx, dy, dz, dx, dy, dz));
    dx = f(dx * inv); dy = f(dy * inv); dz = f(dz * inv);
    intersectCube(P[0], P[1], P[2], dx, dy, dz);
    const tn = vmax(TB.near, 0);
    PH.px = fma(tn, dx, P[0]); PH.py = fma(tn, dy, P[1]); PH.pz = fma(tn, dz, P[2]);
    PH.dx = dx; PH.dy = dy; PH.dz = dz; PH.tx = 1; PH.ty = 1; PH.tz = 1; PH.bounces = 0;
}
function mcmReset(fr, st) {
    const W = fr.width, H = fr.height;
    fr.invW = f(1 / W); fr.invH = f(1 / H);
    for (let j = 0; j < H; j++) {
        for (let i = 0; i < W; i++) {
            const px = pixelNdc(i, W), py = pixelNdc(j, H), k = 4 * (j * W + i);
            RNG.state = hash3(f2u(px), f2u(py), f2u(f(fr.seed)));
            resetPhoton(fr, px, py);
            st[0][k] = PH.px; st[0][k + 1] = PH.py; st[0][k + 2] = PH.pz; st[0][k + 3] = 0;
            st[1][k] = PH.dx; st[1][k + 1] = PH.dy; st[1][k + 2] = PH.dz; st[1][k + 3] = 0;
            st[2][k] = 1; st[2][k + 1] = 1; st[2][k + 2] = 1; st[2][k + 3] = 0;
            st[3][k] = 1; st[3][k + 1] = 1; st[3][k + 2] = 1; st[3][k + 3] = 0;
        }
    }
}
function randomSphere(out) {
    const radius0 = sqrtNr(uniform());
    sincosf(f(f(6.28318530718) * uniform()));
    const ddx = f(radius0 * SC.c), ddy = f(radius0 * SC.s);
    const norm = fma(ddy, ddy, f(ddx * ddx));
    const radius = f(2 * sqrtNr(f(1 - norm)));
    out[0] = f(radius * ddx); out[1] = f(radius * ddy); out[2] = fma(-2, norm, 1);
}
const SPH = new Float32Array(3);
// frame params: { width, height, mvpInv, seed, extinction, anisotropy, bounces, steps } ; y0/y1 optional row range
function mcmIntegrate(sc, fr, st) {
    const W = fr.width, H = fr.height, invExt = f(1 / f(fr.extinction)), g = f(fr.anisotropy);
    fr.invW = f(1 / W); fr.invH = f(1 / H);
    const y0 = fr.y0 || 0, y1 = fr.y1 === undefined ? H : fr.y1;
    for (let j = y0; j < y1; j++) {
        for (let i = 0; i < W; i++) {
            const ndcx = pixelNdc(i, W), ndcy = pixelNdc(j, H), k = 4 * (j * W + i);
            PH.px = st[0][k]; PH.py = st[0][k + 1]; PH.pz = st[0][k + 2];
            PH.dx = st[1][k]; PH.dy = st[1][k + 1]; PH.dz = st[1][k + 2]; PH.bounces = Math.floor(f(st[1][k + 3] + 0.5));
            PH.tx = st[2][k]; PH.ty = st[2][k + 1]; PH.tz = st[2][k + 2];
            let rx = st[3][k], ry = st[3][k + 1], rz = st[3][k + 2], samples = Math.floor(f(st[3][k + 3] + 0.5));
            RNG.state = hash3(f2u(ndcToUv(ndcx)), f2u(ndcToUv(ndcy)), f2u(f(fr.seed)));
            for (let s = 0; s < fr.steps; s++) {
                const dist = exponential(invExt);
                PH.px = fma(dist, PH.dx, PH.px); PH.py = fma(dist, PH.dy, PH.py); PH.pz = fma(dist, PH.dz, PH.pz);
                sampleVolumeColor(sc, PH.px, PH.py, PH.pz);
                const cr = COL[0], cg = COL[1], cb = COL[2], ca = COL[3];
                const pNull = f(1 - ca);
                const pScat = PH.bounces >= fr.bounces ? 0 : f(ca * vmax(vmax(cr, cg), cb));
                const pAbs = f(f(1 - pNull) - pScat);
                const wheel = uniform();
                const oob = PH.px > 1 || PH.py > 1 || PH.pz > 1 || PH.px < 0 || PH.py < 0 || PH.pz < 0;
                if (oob || wheel < pAbs) {
                    let qx = 0, qy = 0, qz = 0;
                    if (oob) { qx = f(PH.tx * sc.env[0]); qy = f(PH.ty * sc.env[1]); qz = f(PH.tz * sc.env[2]); }
                    samples++;
                    const invN = rcpNr(samples);
                    rx = f(rx + f(f(qx - rx) * invN)); ry = f(ry + f(f(qy - ry) * invN)); rz = f(rz + f(f(qz - rz) * invN));
                    resetPhoton(fr, ndcx, ndcy);
                } else if (wheel < f(pAbs + pScat)) {
                    PH.tx = f(PH.tx * cr); PH.ty = f(PH.ty * cg); PH.tz = f(PH.tz * cb);
                    randomSphere(SPH);
                    if (Math.abs(g) < f(1e-5)) {
                        PH.dx = SPH[0]; PH.dy = SPH[1]; PH.dz = SPH[2];
                    } else {
                        const g2 = f(g * g);
                        const c = f(f(1 - g2) * rcpNr(fma(f(2 * g), uniform(), f(1 - g))));
                        const hg = f(fma(-c, c, f(1 + g2)) * rcpNr(f(2 * g)));
                        const ud = dot3(SPH[0], SPH[1], SPH[2], PH.dx, PH.dy, PH.dz);
                        let cx = fma(-ud, PH.dx, SPH[0]), cy = fma(-ud, PH.dy, SPH[1]), cz = fma(-ud, PH.dz, SPH[2]);
                        const inv = rsqrtNr(dot3(cx, cy, cz, cx, cy, cz));
                        cx = f(cx * inv); cy = f(cy * inv); cz = f(cz * inv);
                        const sn = sqrtNr(fma(-hg, hg, 1));
                        const nx = fma(sn, cx, f(hg * PH.dx)), ny = fma(sn, cy, f(hg * PH.dy)), nz = fma(sn, cz, f(hg * PH.dz));
                        PH.dx = nx; PH.dy = ny; PH.dz = nz;
                    }
                    PH.bounces++;
                }
            }
            st[0][k] = PH.px; st[0][k + 1] = PH.py; st[0][k + 2] = PH.pz; st[0][k + 3] = 0;
            st[1][k] = PH.dx; st[1][k + 1] = PH.dy; st[1][k + 2] = PH.dz; st[1][k + 3] = PH.bounces;
            st[2][k] = PH.tx; st[2][k + 1] = PH.ty; st[2][k + 2] = PH.tz; st[2][k + 3] = 0;
            st[3][k] = rx; st[3][k + 1] = ry; st[3][k + 2] = rz; st[3][k + 3] = samples;
        }
    }
}

module.exports = { Scene, mipRender, mcmReset, mcmIntegrate, pcg, hash3, logf, rcpNr, rsqrtNr, SRGB };

// ---- command line: node raymarch.js <job.json>  (volume and matrix come from files written by the caller) -----------
if (require.main === module) {
    const fs = require('fs');
    const job = JSON.parse(fs.readFileSync(process.argv[2]));
    const vol = new Uint8Array(fs.readFileSync(job.volume));
    const sc = new Scene(vol, job.nx, job.ny, job.nz, job.filter || 'linear', null, null);
    const mvpInv = new Float32Array(new Uint32Array(job.mvp_inverse_bits).buffer);
    const W = job.width, H = job.height, out = { kind: job.kind };
    if (job.kind === 'mip') {
        const acc = new Uint8Array(W * H);
        const t0 = process.hrtime.bigint();
        for (let k = 0; k < job.frames; k++) { mipRender(sc, { width: W, height: H, mvpInv, offset: job.offsets[k], steps: job.steps }, acc); }
        out.seconds = Number(process.hrtime.bigint() - t0) * 1e-9;
        if (job.output) { fs.writeFileSync(job.output, Buffer.from(acc.buffer)); }
    } else if (job.kind === 'mcm') {
        const st = [0, 1, 2, 3].map(() => new Float32Array(W * H * 4));
        const fr = { width: W, height: H, mvpInv, seed: job.reset_seed, extinction: job.extinction, anisotropy: job.anisotropy,
                     bounces: job.bounces, steps: job.steps, y0: job.y0, y1: job.y1 };
        mcmReset(fr, st);
        sc.samples = 0;
        const t0 = process.hrtime.bigint();
        for (let k = 0; k < job.seeds.length; k++) { fr.seed = job.seeds[k]; mcmIntegrate(sc, fr, st); }
        out.seconds = Number(process.hrtime.bigint() - t0) * 1e-9;
        if (job.output) { fs.writeFileSync(job.output, Buffer.concat(st.map(a => Buffer.from(a.buffer)))); }
    } else {
        throw new Error('unknown job kind ' + job.kind);
    }
    out.samples = sc.samples;
    console.log(JSON.stringify(out));
}
