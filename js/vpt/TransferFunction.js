'use strict';
// The transfer-function widget of the reference as DATA: the bump list its Save / Load buttons exchange as JSON
// (src/js/ui/TransferFunction/TransferFunction.js:74-85) and the RGBA8 texture its canvas holds for renderer.setTransferFunction
// (render() :110-121, src/glsl/TransferFunction.glsl:32-35).  No DOM: what a user carries over is the bump file; texture() turns
// it into texels on the GPU (vpt_transfer_function_rasterize, include/vpt.h).
const fs = require('fs');
const { native } = require('./native.js');

const num = (v, fallback) => (v === undefined || v === null ? fallback : Number(v));

class TransferFunction {
    constructor(gl, bumps, width, height) {
        this._gl = gl;
        this.transferFunctionWidth = width === undefined ? 256 : width;      // :31-35
        this.transferFunctionHeight = height === undefined ? 256 : height;
        this.bumps = [];
        for (const b of bumps || []) this.addBump(b);
    }

    // the widget's list operations (:127-176)
    addBump(options) {
        const o = options || {}, p = o.position || {}, s = o.size || {}, c = o.color || {};
        this.bumps.push({
            position: { x: num(p.x, 0.5), y: num(p.y, 0.5) },                  // defaults: addBump() :129-144
            size: { x: num(s.x, 0.2), y: num(s.y, 0.2) },
            color: { r: num(c.r, 1), g: num(c.g, 0), b: num(c.b, 0), a: num(c.a, 1) },
        });
        return this.bumps.length - 1;
    }
    removeBump(index) { this.bumps.splice(index, 1); }
    removeAllBumps() { this.bumps = []; }
    resizeTransferFunction(width, height) { this.transferFunctionWidth = width; this.transferFunctionHeight = height; }

    // Save / Load (:74-85): the bump array as JSON, nothing else
    dumps() { return JSON.stringify(this.bumps); }
    loads(text) {
        const data = JSON.parse(text);
        if (!Array.isArray(data)) throw new TypeError('a transfer-function file is a JSON array of bumps');
        this.bumps = [];
        for (const b of data) this.addBump(b);
        return this;
    }
    save(path) { fs.writeFileSync(path, this.dumps()); }
    load(path) { return this.loads(fs.readFileSync(path, 'utf8')); }

    // [count][8] float32: position.xy, size.xy, color.rgba (struct vpt_tf_bump)
    packed() {
        const a = new Float32Array(this.bumps.length * 8);
        this.bumps.forEach((b, k) => a.set([b.position.x, b.position.y, b.size.x, b.size.y, b.color.r, b.color.g, b.color.b, b.color.a], 8 * k));
        return a;
    }
    // { width, height, data: Uint8Array [height][width][4] } for renderer.setTransferFunction: row 0 = the canvas's top row
    // (position.y = 1), as texImage2D(canvas) transfers it; unpremultiply (default true): the colour divided by alpha again, as a
    // browser hands a premultiplied WebGL canvas over
    texture(unpremultiply) {
        const w = this.transferFunctionWidth, h = this.transferFunctionHeight;
        const data = new Uint8Array(w * h * 4);
        native().transferFunctionRasterize(this._gl._h, this.packed(), w, h, unpremultiply === false ? 0 : 1, data);
        return { width: w, height: h, data };
    }
    get value() { return this.texture(); }                                     // `get value()` :123-125
}
module.exports = { TransferFunction };
