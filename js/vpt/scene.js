'use strict';
// Node / Component / Transform / PerspectiveCamera — src/js/Node.js:3-51, Component.js, Transform.js:4-116,
// PerspectiveCamera.js:4-19 — and the inverse-MVP recipe of MIPRenderer.js:86-97.
const { EventTarget, Event } = require('./EventTarget.js');
const { vec3, quat, mat4 } = require('./math.js');

class Component extends EventTarget {
    constructor(node) { super(); this.node = node; }
}

class Transform extends Component {

constructor(node) {
    super(node);
    this.version = 0;
    this._localRotation = quat.create();
    this._localTranslation = vec3.create();
    this._localScale = new Float32Array([1, 1, 1]);
}

get localRotation() { return quat.clone(this._localRotation); }
get localTranslation() { return vec3.clone(this._localTranslation); }
get localScale() { return vec3.clone(this._localScale); }

set localRotation(v) { this._localRotation = quat.clone(v); this.version++; this.dispatchEvent(new Event('change')); }
set localTranslation(v) { this._localTranslation = vec3.clone(v); this.version++; this.dispatchEvent(new Event('change')); }
set localScale(v) { this._localScale = vec3.clone(v); this.version++; this.dispatchEvent(new Event('change')); }

get localMatrix() {
    return mat4.fromRotationTranslationScale(mat4.create(), this._localRotation, this._localTranslation, this._localScale);
}

get globalTranslation() {                                                          // Transform.js:35-37
    const m = this.globalMatrix, out = vec3.create();
    out[0] = m[12]; out[1] = m[13]; out[2] = m[14];
    return out;
}

get globalMatrix() {
    const parent = this.node ? this.node.parent : null;
    if (parent) {
        const globalMatrix = parent.transform.globalMatrix;
        return mat4.multiply(globalMatrix, globalMatrix, this.localMatrix);
    }
    return this.localMatrix;
}

get inverseLocalMatrix() {
    const matrix = mat4.create();
    mat4.scale(matrix, matrix, vec3.inverse(vec3.create(), this._localScale));
    mat4.multiply(matrix, matrix, mat4.fromQuat(mat4.create(), quat.invert(quat.create(), this._localRotation)));
    mat4.translate(matrix, matrix, vec3.negate(vec3.create(), this._localTranslation));
    return matrix;
}

get inverseGlobalMatrix() {
    const parent = this.node ? this.node.parent : null;
    if (parent) {
        const inverseGlobalMatrix = parent.transform.inverseGlobalMatrix;
        return mat4.multiply(inverseGlobalMatrix, this.inverseLocalMatrix, inverseGlobalMatrix);
    }
    return this.inverseLocalMatrix;
}

}

class Node {

constructor() {
    this.children = [];
    this.parent = null;
    this.components = [new Transform(this)];
}

addChild(node) {
    if (node.parent) { node.parent.removeChild(node); }
    this.children.push(node);
    node.parent = this;
}

removeChild(node) {
    const index = this.children.indexOf(node);
    if (index >= 0) { this.children.splice(index, 1); node.parent = null; }
}

getComponent(type) { return this.components.find(component => component instanceof type); }

get transform() { return this.getComponent(Transform); }

}

class PerspectiveCamera extends Component {

constructor(node, options) {
    super(node);
    options = options || {};
    this.fovy = options.fovy !== undefined ? options.fovy : 1;
    this.aspect = options.aspect !== undefined ? options.aspect : 1;
    this.near = options.near !== undefined ? options.near : 0.1;
    this.far = options.far !== undefined ? options.far : 100;
}

get projectionMatrix() { return mat4.perspective(mat4.create(), this.fovy, this.aspect, this.near, this.far); }

}

// RenderingContext.js:38-40,121
function defaultCamera(aspect) {
    const node = new Node();
    node.transform.localTranslation = [0, 0, 2];
    const camera = new PerspectiveCamera(node);
    camera.aspect = aspect === undefined ? 1 : aspect;
    node.components.push(camera);
    return node;
}

// MIPRenderer.js:86-97 (= EAMRenderer.js:105-116, MCSRenderer.js:93-104, MCMRenderer.js:95-106,164-175)
function mvpInverseMatrix(camera, volumeTransform) {
    const centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]);
    const modelMatrix = volumeTransform.globalMatrix;
    const viewMatrix = camera.transform.inverseGlobalMatrix;
    const projectionMatrix = camera.getComponent(PerspectiveCamera).projectionMatrix;
    const matrix = mat4.create();
    mat4.multiply(matrix, centerMatrix, matrix);
    mat4.multiply(matrix, modelMatrix, matrix);
    mat4.multiply(matrix, viewMatrix, matrix);
    mat4.multiply(matrix, projectionMatrix, matrix);
    mat4.invert(matrix, matrix);
    return matrix;
}

// ISORenderer.js:152-166: the light direction (view space) taken into model space and normalised
function isoLightDirection(camera, volumeTransform, light) {
    const centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5]);
    const modelMatrix = volumeTransform.globalMatrix;
    const viewMatrix = camera.transform.inverseGlobalMatrix;
    const matrix = mat4.create();
    mat4.multiply(matrix, centerMatrix, matrix);
    mat4.multiply(matrix, modelMatrix, matrix);
    mat4.multiply(matrix, viewMatrix, matrix);
    mat4.invert(matrix, matrix);
    const out = vec3.transformMat4(vec3.create(), light, matrix);
    vec3.normalize(out, out);
    return out;
}

module.exports = { Component, Transform, Node, PerspectiveCamera, defaultCamera, mvpInverseMatrix, isoLightDirection };
