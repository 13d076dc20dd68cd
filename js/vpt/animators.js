'use strict';
// src/js/animators/CircleAnimator.js:3-42 — the time-driven camera animator the reference's animation recorder steps
// (RenderingContext.js:283-284).  Pinned by tests/golden/circle_animator_r01.json (the reference's CircleAnimator run
// under node).  OrbitCameraAnimator (SURVEY section 8f row 4) is the reference's interactive camera, headless here: the same
// state, handlers and arithmetic, fed with event-like objects and an explicit clock instead of DOM events and Date.now();
// pinned by tests/golden/orbit_animator_r01.json (the reference's class run under node on scripted input).
const { mat4, quat, vec3 } = require('./math.js');

class CircleAnimator {

constructor(node, options) {
    this.node = node;
    Object.assign(this, { center: [0, 0, 0], direction: [1, 0, 0], radius: 1, frequency: 1 }, options || {});
}

update(t) {                                                                        // :16-40
    const scale = mat4.fromScaling(mat4.create(), [this.radius, this.radius, this.radius]);
    const angle = this.frequency * t * 2 * Math.PI;
    const phase = mat4.fromRotation(mat4.create(), angle, [0, 0, 1]);
    const from = [0, 0, 1];
    const to = vec3.normalize(vec3.create(), this.direction);
    const axis = vec3.cross(vec3.create(), from, to);
    const slant = vec3.dot(from, to);
    const orientationQuat = [axis[0], axis[1], axis[2], slant];
    const orientation = mat4.fromQuat(mat4.create(), orientationQuat);
    const translation = mat4.fromTranslation(mat4.create(), this.center);
    const composite = mat4.create();
    mat4.multiply(composite, composite, translation);
    mat4.multiply(composite, composite, orientation);
    mat4.multiply(composite, composite, phase);
    mat4.multiply(composite, composite, scale);
    const position = [1, 0, 0];
    this.node.transform.localTranslation = vec3.transformMat4(position, position, composite);
    this.node.transform.localRotation = orientationQuat;
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
