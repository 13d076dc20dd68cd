'use strict';
// Multi-GPU frame gather for the Node.js host (no reference counterpart; SURVEY section 8e): one process per GPU, the
// renderer sharded with options.shard, every frame all-gathered over RCCL/xGMI by the pipeline below the C ABI
// (vpt_gather_*).  The 128-byte id from FrameGather.uniqueId() is created by ONE rank and handed to the others by the
// launcher (IPC message, file, environment) before they construct their FrameGather.
const { native } = require('./native.js');

class FrameGather {

// root: the display rank that receives every frame (grouped send/recv), or -1 = all ranks (all_gather, the default)
constructor(renderer, id, rank, world, root) {
    this._renderer = renderer;
    this._h = native().gatherCreate(renderer._h, id, rank, world);
    if (root !== undefined && root !== -1) { native().gatherSetRoot(this._h, root); }
}

static uniqueId() { return native().gatherUniqueId(); }

// one renderer.render() into the next send buffer + asynchronous all_gather; returns immediately
render() {
    const r = this._renderer;
    r._bindVolume();
    native().gatherRender(this._h, r._prepareFused());
}

synchronize() { native().gatherSynchronize(this._h); }

// the most recently gathered frame, rows in order: { data: Uint16Array RGBA16F bits, width, height }
getFrame() {
    const size = this._renderer._size();
    const out = new Uint16Array(size[0] * size[1] * 4);
    native().gatherReadFrame(this._h, out);
    return { data: out, width: size[0], height: size[1], format: 'RGBA16F' };
}

destroy() { if (this._h) { native().gatherDestroy(this._h); this._h = null; } }

}

module.exports = { FrameGather };
