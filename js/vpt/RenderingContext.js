'use strict';
// src/js/RenderingContext.js:20-229, headless: the caller of the renderer path (SURVEY section 8b "Caller to
// reproduce").  Same members and methods minus the browser parts (canvas, WebGL context loss, animators, recording):
// where the reference blits the tone mapper's texture to the canvas (:199-209), getFrame() reads it back.
//   new RenderingContext({ resolution, filter, device, rng })     resolution: number or { width, height }
const { EventTarget, CustomEvent } = require('./EventTarget.js');
const { Context } = require('./Context.js');
const { OrbitCameraAnimator } = require('./animators.js');
const { Node, Transform, PerspectiveCamera } = require('./scene.js');
const { Volume } = require('./Volume.js');
const { RendererFactory } = require('./renderers/RendererFactory.js');
const { ToneMapperFactory } = require('./tonemappers/ToneMapperFactory.js');

class RenderingContext extends EventTarget {

constructor(options) {
    super();
    options = options || {};
    this.render = this.render.bind(this);
    this.gl = new Context(options.device || 0);                                   // initGL(), :61-105
    this.environmentTexture = { data: new Uint8Array([255, 255, 255, 255]), width: 1, height: 1 };   // :90-101
    this._rng = options.rng;
    this._resolution = options.resolution !== undefined ? options.resolution : 512;   // :35
    this.filter = options.filter !== undefined ? options.filter : 'linear';           // :36
    this.camera = new Node();                                                         // :38-40
    this.camera.transform.localTranslation = [0, 0, 2];
    this.camera.components.push(new PerspectiveCamera(this.camera));
    this.camera.transform.addEventListener('change', () => {                          // :42-46
        if (this.renderer) { this.renderer.reset(); }
    });
    this.volume = new Volume(this.gl);                                                // :56
    this.volumeTransform = new Transform(new Node());                                 // :57
    this.renderer = null;
    this.toneMapper = null;
    this.cameraAnimator = new OrbitCameraAnimator(this.camera, null);                 // :54 (headless: no canvas to listen on)
    const size = this._size();
    this.resize(size[0], size[1]);
}

_size() {
    const r = this._resolution;
    return typeof r === 'number' ? [r, r] : [r.width, r.height];
}

destroy() {
    if (this.toneMapper) { this.toneMapper.destroy(); this.toneMapper = null; }
    if (this.renderer) { this.renderer.destroy(); this.renderer = null; }
    if (this.volume) { this.volume.destroy(); }
    this.gl.destroy();
}

resize(width, height) {                                                               // :117-121
    this.camera.getComponent(PerspectiveCamera).aspect = width / height;
}

async setVolume(reader) {                                                             // :123-133
    const old = this.volume;
    this.volume = new Volume(this.gl, reader);
    this.volume.addEventListener('progress', e => {
        this.dispatchEvent(new CustomEvent('progress', { detail: e.detail }));
    });
    await this.volume.load();
    this.volume.setFilter(this.filter);
    if (this.renderer) { this.renderer.setVolume(this.volume); }
    if (old) { old.destroy(); }                                                        // device memory is not garbage-collected
}

setEnvironmentMap(image) {                                                            // :135-140 — { data: RGBA8, width, height }
    this.environmentTexture = image;
    if (this.renderer) { this.renderer.setEnvironmentMap(image); }
}

setFilter(filter) {                                                                   // :142-150
    this.filter = filter;
    if (this.volume) {
        this.volume.setFilter(filter);
        if (this.renderer) { this.renderer.reset(); }
    }
}

chooseRenderer(renderer) {                                                            // :152-167
    if (this.renderer) { this.renderer.destroy(); }
    const rendererClass = RendererFactory(renderer);
    const options = { resolution: this._resolution, transform: this.volumeTransform };
    if (this._rng) { options.rng = this._rng; }
    this.renderer = new rendererClass(this.gl, this.volume, this.camera, this.environmentTexture, options);
    this.renderer.reset();
    if (this.toneMapper) { this.toneMapper.setTexture(this.renderer); }
    this.isTransformationDirty = true;
}

chooseToneMapper(toneMapper) {                                                        // :169-188
    if (this.toneMapper) { this.toneMapper.destroy(); }
    const toneMapperClass = ToneMapperFactory(toneMapper);
    this.toneMapper = new toneMapperClass(this.gl, this.renderer || null, { resolution: this._resolution });
}

render() {                                                                            // :190-210
    if (!this.renderer || !this.toneMapper) { return; }
    this.renderer.render();
    this.toneMapper.render();
}

// what the reference puts on the canvas: the tone mapper's RGBA8 image, read back
getFrame() { return this.toneMapper.getTexture(); }

// :259-305, headless and deterministic: for every frame time t = startTime + i / fps the camera animator is stepped, the
// renderer reset and `passes` render() calls made (the reference renders for `frameTime` seconds of wall clock), and the
// tone-mapped frame is written as directory/frameNNNN.png.  options: { directory, startTime, endTime, fps, passes }
recordAnimationToImageSequence(options) {
    const fs = require('fs'), path = require('path');
    const { encodePNG } = require('./png.js');
    options = options || {};
    if (!this.cameraAnimator || !this.renderer || !this.toneMapper) {
        throw new Error('recordAnimationToImageSequence needs a cameraAnimator, a renderer and a tone mapper');
    }
    const startTime = options.startTime || 0, endTime = options.endTime !== undefined ? options.endTime : 1;
    const fps = options.fps || 30, passes = options.passes || 16;
    const frames = Math.max(Math.ceil((endTime - startTime) * fps), 1);               // :261
    const timeStep = 1 / fps;
    fs.mkdirSync(options.directory, { recursive: true });
    const files = [];
    for (let i = 0; i < frames; i++) {
        const t = startTime + i * timeStep;                                           // :283
        this.cameraAnimator.update(t);
        this.renderer.reset();                                                        // :286
        for (let k = 0; k < passes; k++) { this.render(); }
        const file = path.join(options.directory, 'frame' + String(i).padStart(4, '0') + '.png');   // :291
        fs.writeFileSync(file, encodePNG(this.getFrame(), true));
        files.push(file);
        this.dispatchEvent(new CustomEvent('animationprogress', { detail: (i + 1) / frames }));     // :298-300
    }
    return files;
}

get resolution() { return this._resolution; }                                         // :212-214

set resolution(resolution) {                                                          // :216-229
    this._resolution = resolution;
    if (this.renderer) { this.renderer.setResolution(resolution); }
    if (this.toneMapper) {
        this.toneMapper.setResolution(resolution);
        if (this.renderer) { this.toneMapper.setTexture(this.renderer); }
    }
}

}
module.exports = { RenderingContext };
