'use strict';
// mat4 / quat / vec3: restatement of the published gl-matrix 3.4.1 algorithms the path uses (Float32Array storage,
// double arithmetic per element) — the reference vendors gl-matrix (src/lib/gl-matrix-min.js) and calls exactly these
// functions in MIPRenderer.js:86-97, Transform.js:26-29,64-70, PerspectiveCamera.js:15-17.  Pinned bit-for-bit by
// tests/golden/mvp_inverse.json (js/test/test_host.js).
const vec3 = {
    create() { return new Float32Array(3); },
    clone(a) { const o = new Float32Array(3); o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; return o; },
    negate(out, a) { out[0] = -a[0]; out[1] = -a[1]; out[2] = -a[2]; return out; },
    inverse(out, a) { out[0] = 1.0 / a[0]; out[1] = 1.0 / a[1]; out[2] = 1.0 / a[2]; return out; },
    transformMat4(out, a, m) {
        const x = a[0], y = a[1], z = a[2];
        let w = m[3] * x + m[7] * y + m[11] * z + m[15];
        w = w || 1.0;
        out[0] = (m[0] * x + m[4] * y + m[8] * z + m[12]) / w;
        out[1] = (m[1] * x + m[5] * y + m[9] * z + m[13]) / w;
        out[2] = (m[2] * x + m[6] * y + m[10] * z + m[14]) / w;
        return out;
    },
    add(out, a, b) { out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2]; return out; },
    distance(a, b) { return Math.hypot(b[0] - a[0], b[1] - a[1], b[2] - a[2]); },
    transformQuat(out, a, q) {
        const qx = q[0], qy = q[1], qz = q[2], qw = q[3], x = a[0], y = a[1], z = a[2];
        let uvx = qy * z - qz * y, uvy = qz * x - qx * z, uvz = qx * y - qy * x;
        let uuvx = qy * uvz - qz * uvy, uuvy = qz * uvx - qx * uvz, uuvz = qx * uvy - qy * uvx;
        const w2 = qw * 2;
        uvx *= w2; uvy *= w2; uvz *= w2;
        uuvx *= 2; uuvy *= 2; uuvz *= 2;
        out[0] = x + uvx + uuvx; out[1] = y + uvy + uuvy; out[2] = z + uvz + uuvz;
        return out;
    },
    cross(out, a, b) {
        const ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2];
        out[0] = ay * bz - az * by; out[1] = az * bx - ax * bz; out[2] = ax * by - ay * bx;
        return out;
    },
    dot(a, b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; },
    normalize(out, a) {
        const x = a[0], y = a[1], z = a[2];
        let len = x * x + y * y + z * z;
        if (len > 0) { len = 1 / Math.sqrt(len); }
        out[0] = a[0] * len; out[1] = a[1] * len; out[2] = a[2] * len;
        return out;
    },
};
const quat = {
    create() { const o = new Float32Array(4); o[3] = 1; return o; },
    clone(a) { const o = new Float32Array(4); o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; return o; },
    invert(out, a) {
        const a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        const dot = a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        const invDot = dot ? 1.0 / dot : 0;
        out[0] = -a0 * invDot; out[1] = -a1 * invDot; out[2] = -a2 * invDot; out[3] = a3 * invDot;
        return out;
    },
    rotateX(out, a, rad) {
        rad *= 0.5;
        const ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = Math.sin(rad), bw = Math.cos(rad);
        out[0] = ax * bw + aw * bx; out[1] = ay * bw + az * bx; out[2] = az * bw - ay * bx; out[3] = aw * bw - ax * bx;
        return out;
    },
    rotateY(out, a, rad) {
        rad *= 0.5;
        const ax = a[0], ay = a[1], az = a[2], aw = a[3], by = Math.sin(rad), bw = Math.cos(rad);
        out[0] = ax * bw - az * by; out[1] = ay * bw + aw * by; out[2] = az * bw + ax * by; out[3] = aw * bw - ay * by;
        return out;
    },
    setAxisAngle(out, axis, rad) {
        rad = rad * 0.5;
        const s = Math.sin(rad);
        out[0] = s * axis[0]; out[1] = s * axis[1]; out[2] = s * axis[2]; out[3] = Math.cos(rad);
        return out;
    },
    multiply(out, a, b) {
        const ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
        out[0] = ax * bw + aw * bx + ay * bz - az * by;
        out[1] = ay * bw + aw * by + az * bx - ax * bz;
        out[2] = az * bw + aw * bz + ax * by - ay * bx;
        out[3] = aw * bw - ax * bx - ay * by - az * bz;
        return out;
    },
};
const mat4 = {
    create() { const o = new Float32Array(16); o[0] = 1; o[5] = 1; o[10] = 1; o[15] = 1; return o; },
    fromTranslation(out, v) {
        out.fill(0); out[0] = 1; out[5] = 1; out[10] = 1; out[15] = 1;
        out[12] = v[0]; out[13] = v[1]; out[14] = v[2];
        return out;
    },
    fromScaling(out, v) {
        out.fill(0); out[0] = v[0]; out[5] = v[1]; out[10] = v[2]; out[15] = 1;
        return out;
    },
    fromRotation(out, rad, axis) {
        let x = axis[0], y = axis[1], z = axis[2];
        let len = Math.hypot(x, y, z);
        if (len < 0.000001) { return null; }
        len = 1 / len; x *= len; y *= len; z *= len;
        const s = Math.sin(rad), c = Math.cos(rad), t = 1 - c;
        out[0] = x * x * t + c; out[1] = y * x * t + z * s; out[2] = z * x * t - y * s; out[3] = 0;
        out[4] = x * y * t - z * s; out[5] = y * y * t + c; out[6] = z * y * t + x * s; out[7] = 0;
        out[8] = x * z * t + y * s; out[9] = y * z * t - x * s; out[10] = z * z * t + c; out[11] = 0;
        out[12] = 0; out[13] = 0; out[14] = 0; out[15] = 1;
        return out;
    },
    multiply(out, a, b) {
        const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
        const a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
        const bb = Array.prototype.slice.call(b);
        for (let c = 0; c < 4; c++) {
            const b0 = bb[4 * c], b1 = bb[4 * c + 1], b2 = bb[4 * c + 2], b3 = bb[4 * c + 3];
            out[4 * c] = b0 * a00 + b1 * a10 + b2 * a20 + b3 * a30;
            out[4 * c + 1] = b0 * a01 + b1 * a11 + b2 * a21 + b3 * a31;
            out[4 * c + 2] = b0 * a02 + b1 * a12 + b2 * a22 + b3 * a32;
            out[4 * c + 3] = b0 * a03 + b1 * a13 + b2 * a23 + b3 * a33;
        }
        return out;
    },
    invert(out, a) {
        const a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
        const a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11], a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
        const b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
        const b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
        const b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
        const b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
        let det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
        if (!det) { return null; }
        det = 1.0 / det;
        out[0] = (a11 * b11 - a12 * b10 + a13 * b09) * det;
        out[1] = (a02 * b10 - a01 * b11 - a03 * b09) * det;
        out[2] = (a31 * b05 - a32 * b04 + a33 * b03) * det;
        out[3] = (a22 * b04 - a21 * b05 - a23 * b03) * det;
        out[4] = (a12 * b08 - a10 * b11 - a13 * b07) * det;
        out[5] = (a00 * b11 - a02 * b08 + a03 * b07) * det;
        out[6] = (a32 * b02 - a30 * b05 - a33 * b01) * det;
        out[7] = (a20 * b05 - a22 * b02 + a23 * b01) * det;
        out[8] = (a10 * b10 - a11 * b08 + a13 * b06) * det;
        out[9] = (a01 * b08 - a00 * b10 - a03 * b06) * det;
        out[10] = (a30 * b04 - a31 * b02 + a33 * b00) * det;
        out[11] = (a21 * b02 - a20 * b04 - a23 * b00) * det;
        out[12] = (a11 * b07 - a10 * b09 - a12 * b06) * det;
        out[13] = (a00 * b09 - a01 * b07 + a02 * b06) * det;
        out[14] = (a31 * b01 - a30 * b03 - a32 * b00) * det;
        out[15] = (a20 * b03 - a21 * b01 + a22 * b00) * det;
        return out;
    },
    perspective(out, fovy, aspect, near, far) {
        const f = 1.0 / Math.tan(fovy / 2);
        out.fill(0);
        out[0] = f / aspect; out[5] = f; out[11] = -1;
        if (far != null && far !== Infinity) {
            const nf = 1 / (near - far);
            out[10] = (far + near) * nf;
            out[14] = 2 * far * near * nf;
        } else {
            out[10] = -1; out[14] = -2 * near;
        }
        return out;
    },
    fromRotationTranslationScale(out, q, v, s) {
        const x = q[0], y = q[1], z = q[2], w = q[3];
        const x2 = x + x, y2 = y + y, z2 = z + z;
        const xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
        const wx = w * x2, wy = w * y2, wz = w * z2;
        const sx = s[0], sy = s[1], sz = s[2];
        out[0] = (1 - (yy + zz)) * sx; out[1] = (xy + wz) * sx; out[2] = (xz - wy) * sx; out[3] = 0;
        out[4] = (xy - wz) * sy; out[5] = (1 - (xx + zz)) * sy; out[6] = (yz + wx) * sy; out[7] = 0;
        out[8] = (xz + wy) * sz; out[9] = (yz - wx) * sz; out[10] = (1 - (xx + yy)) * sz; out[11] = 0;
        out[12] = v[0]; out[13] = v[1]; out[14] = v[2]; out[15] = 1;
        return out;
    },
    scale(out, a, v) {
        const x = v[0], y = v[1], z = v[2];
        for (let i = 0; i < 4; i++) { out[i] = a[i] * x; out[4 + i] = a[4 + i] * y; out[8 + i] = a[8 + i] * z; out[12 + i] = a[12 + i]; }
        return out;
    },
    fromQuat(out, q) {
        const x = q[0], y = q[1], z = q[2], w = q[3];
        const x2 = x + x, y2 = y + y, z2 = z + z;
        const xx = x * x2, yx = y * x2, yy = y * y2, zx = z * x2, zy = z * y2, zz = z * z2;
        const wx = w * x2, wy = w * y2, wz = w * z2;
        out[0] = 1 - yy - zz; out[1] = yx + wz; out[2] = zx - wy; out[3] = 0;
        out[4] = yx - wz; out[5] = 1 - xx - zz; out[6] = zy + wx; out[7] = 0;
        out[8] = zx + wy; out[9] = zy - wx; out[10] = 1 - xx - yy; out[11] = 0;
        out[12] = 0; out[13] = 0; out[14] = 0; out[15] = 1;
        return out;
    },
    translate(out, a, v) {
        const x = v[0], y = v[1], z = v[2];
        const a0 = Array.prototype.slice.call(a);
        for (let i = 0; i < 12; i++) { out[i] = a0[i]; }
        out[12] = a0[0] * x + a0[4] * y + a0[8] * z + a0[12];
        out[13] = a0[1] * x + a0[5] * y + a0[9] * z + a0[13];
        out[14] = a0[2] * x + a0[6] * y + a0[10] * z + a0[14];
        out[15] = a0[3] * x + a0[7] * y + a0[11] * z + a0[15];
        return out;
    },
};
module.exports = { vec3, quat, mat4 };
