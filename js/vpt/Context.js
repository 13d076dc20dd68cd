'use strict';
// Device context: the object passed where the reference passes its WebGL2RenderingContext (`gl`,
// RenderingContext.js:66-106).
const { native } = require('./native.js');

class Context {
    constructor(device) {
        this.device = device === undefined ? 0 : device;
        this._h = native().contextCreate(this.device);
    }
    static deviceCount() { return native().deviceCount(); }
    synchronize() { native().contextSynchronize(this._h); }
    destroy() { if (this._h) { native().contextDestroy(this._h); this._h = null; } }
}
module.exports = { Context };
