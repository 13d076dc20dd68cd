'use strict';
// src/js/animators/CircleAnimator.js:3-42 — the time-driven camera animator the reference's animation recorder steps
// (RenderingContext.js:283-284).  Pinned by tests/golden/circle_animator_r01.json (the reference's CircleAnimator run
// under node).  OrbitCameraAnimator (SURVEY section 8f row 4) is the reference's interactive camera, headless here: the same
// state, handlers and arithmetic, fed with event-like objects and an explicit clock instead of DOM events and Date.now();
// pinned by tests/golden/orbit_animator_r01.json (the reference's class run under node on scripted input).
const { mat4, quat, vec3 } = require('./math.js');

// The pose is what the fixture pins: a point on a circle of `radius` about `center`, in the plane whose normal is `direction`,
// `frequency` turns per unit of t — reached through the SAME chain of 4x4 Float32Array products the reference's update() forms
// (shift * tilt * spin * scale, each product rounded to float32), because its last bits depend on that chain.
class CircleAnimator {

constructor(node, options) {
    this.node = node;
    this.center = [0, 0, 0];
    this.direction = [1, 0, 0];
    this.radius = 1;
    this.frequency = 1;
    const given = options || {};
    for (const key of Object.keys(given)) this[key] = given[key];
}

// the rotation taking +z onto `direction`, the reference's way: the UN-normalised quaternion (z x d, z . d) — also what the node's
// localRotation receives
_tilt() {
    const z = [0, 0, 1];
    const d = vec3.normalize(vec3.create(), this.direction);
    const c = vec3.cross(vec3.create(), z, d);
    return [c[0], c[1], c[2], vec3.dot(z, d)];
}

update(t) {
    const tilt = this._tilt();
    const r = this.radius;
    const factors = [
        mat4.fromTranslation(mat4.create(), this.center),
        mat4.fromQuat(mat4.create(), tilt),
        mat4.fromRotation(mat4.create(), this.frequency * t * 2 * Math.PI, [0, 0, 1]),
        mat4.fromScaling(mat4.create(), [r, r, r]),
    ];
    const pose = factors.reduce((acc, f) => mat4.multiply(acc, acc, f), mat4.create());
    const start = [1, 0, 0];                                                       // a plain array: transformMat4 works in double on it
    const transform = this.node.transform;
    transform.localTranslation = vec3.transformMat4(start, start, pose);
    transform.localRotation = tilt;
}

}
// src/js/animators/OrbitCameraAnimator.js:4-200.  `domElement` is kept for signature parity and never touched; `options.now`
// is the clock _update() reads in milliseconds (default Date.now).
class OrbitCameraAnimator {

constructor(camera, domElement, options) {
    Object.assign(this, { rotationSpeed: 0.005, translationSpeed: 0.005, moveSpeed: 0.001, zoomSpeed: 0.001, now: Date.now }, options || {});   // :14-19
    this._camera = camera;
    this._domElement = domElement;
    this._focus = [0, 0, 0];
    this._focusDistance = vec3.distance(this._focus, this._camera.transform.globalTranslation);   // :25
    this._yaw = 0;
    this._pitch = 0;
    this._forward = false; this._backward = false; this._left = false; this._right = false;
    this._isTranslating = false;
    this._isRotating = false;
    this._time = this.now();
}

_handlePointerDown(e) {                                                            // :50-60
    if (e.button === 0) { this._isRotating = true; } else if (e.button === 1) { this._isTranslating = true; }
}
_handlePointerUp() { this._isTranslating = false; this._isRotating = false; }      // :62-69
_handlePointerMove(e) {                                                            // :71-95
    const dx = e.movementX, dy = e.movementY;
    if (this._isRotating) {
        this._rotateAroundFocus(-dx * this.rotationSpeed, -dy * this.rotationSpeed);   // both branches of the shiftKey test do this
    }
    if (this._isTranslating) {
        // :88-92 multiplies a number by the focus ARRAY (NaN) and hands _move a number: vec3.transformQuat then throws a
        // TypeError in the reference (strict mode, property store on a primitive) before anything is modified
        throw new TypeError("Cannot create property '0' on number 'NaN'");
    }
}
_handleWheel(e) { this._zoom(e.deltaY * this.zoomSpeed); }                         // :97-99
_handleKeyDown(e) { this._key(e, true); }                                          // :101-108
_handleKeyUp(e) { this._key(e, false); }                                           // :110-117
_key(e, down) {
    switch (e.key.toLowerCase()) {
        case 'w': this._forward = down; break;
        case 'a': this._left = down; break;
        case 's': this._backward = down; break;
        case 'd': this._right = down; break;
    }
}

_updateCamera() {                                                                  // :119-131
    const transform = this._camera.transform;
    const rotation = quat.create();
    quat.rotateY(rotation, rotation, this._yaw);
    quat.rotateX(rotation, rotation, this._pitch);
    const translation = vec3.transformQuat(vec3.create(), [0, 0, this._focusDistance], rotation);
    transform.localRotation = rotation;
    transform.localTranslation = vec3.add(vec3.create(), this._focus, translation);
}
_rotateAroundFocus(dx, dy) {                                                       // :133-144
    const twopi = Math.PI * 2, halfpi = Math.PI / 2;
    this._pitch += dy;
    this._pitch = Math.min(Math.max(this._pitch, -halfpi), halfpi);
    this._yaw += dx;
    this._yaw = ((this._yaw % twopi) + twopi) % twopi;
    this._updateCamera();
}
_move(v) {                                                                         // :146-153
    const rotation = quat.create();
    quat.rotateY(rotation, rotation, this._yaw);
    quat.rotateX(rotation, rotation, this._pitch);
    vec3.transformQuat(v, v, rotation);
    vec3.add(this._focus, this._focus, v);
    this._updateCamera();
}
_zoom(amount) { this._focusDistance *= Math.exp(amount); this._updateCamera(); }    // :155-158
_update() {                                                                        // :160-186 (the Ticker callback)
    const t = this.now(), dt = t - this._time;
    this._time = t;
    let dx = 0, dz = 0;
    if (this._forward) { dz -= this.moveSpeed * this._focusDistance * dt; }
    if (this._backward) { dz += this.moveSpeed * this._focusDistance * dt; }
    if (this._left) { dx -= this.moveSpeed * this._focusDistance * dt; }
    if (this._right) { dx += this.moveSpeed * this._focusDistance * dt; }
    if (dx !== 0 || dz !== 0) { this._move([dx, 0, dz]); }
}
update(t) {}                                                                       // :188-190: responds to input only

}
module.exports = { CircleAnimator, OrbitCameraAnimator };
