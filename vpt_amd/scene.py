"""Scene-side inputs of the renderer path: Node / Transform / PerspectiveCamera and the matrix recipe.

Mirrors (names and behaviour) src/js/Node.js:3-51, src/js/Transform.js:4-178, src/js/Component.js,
src/js/PerspectiveCamera.js:4-19 of the reference.  The reference computes these with its vendored
gl-matrix 3.4.1 (Float32Array storage, double-precision arithmetic per element); ``mat4``/``quat``/``vec3``
below restate the published gl-matrix algorithms with exactly that storage/rounding model so that
``uMvpInverseMatrix`` is bit-identical (pinned by tests/golden/mvp_inverse.json).
"""
import math

import numpy as np

from .property_bag import EventTarget, Event


def _f32(values):
    return np.asarray(values, dtype=np.float32)


class vec3:
    @staticmethod
    def create():
        return np.zeros(3, dtype=np.float32)

    @staticmethod
    def clone(a):
        return _f32([float(a[0]), float(a[1]), float(a[2])])

    @staticmethod
    def negate(out, a):
        out[0], out[1], out[2] = -float(a[0]), -float(a[1]), -float(a[2])
        return out

    @staticmethod
    def inverse(out, a):
        out[0], out[1], out[2] = 1.0 / float(a[0]), 1.0 / float(a[1]), 1.0 / float(a[2])
        return out

    @staticmethod
    def transformMat4(out, a, m):
        """gl-matrix 3.4.1 vec3.transformMat4: homogeneous transform with w = 1, divided by the resulting w (or 1 if 0)"""
        x, y, z = float(a[0]), float(a[1]), float(a[2])
        m = [float(v) for v in m]
        w = m[3] * x + m[7] * y + m[11] * z + m[15]
        w = w or 1.0
        out[0] = (m[0] * x + m[4] * y + m[8] * z + m[12]) / w
        out[1] = (m[1] * x + m[5] * y + m[9] * z + m[13]) / w
        out[2] = (m[2] * x + m[6] * y + m[10] * z + m[14]) / w
        return out

    @staticmethod
    def add(out, a, b):
        out[0], out[1], out[2] = float(a[0]) + float(b[0]), float(a[1]) + float(b[1]), float(a[2]) + float(b[2])
        return out

    @staticmethod
    def distance(a, b):
        """gl-matrix 3.4.1 vec3.distance: Math.hypot of the differences"""
        return math.hypot(float(b[0]) - float(a[0]), float(b[1]) - float(a[1]), float(b[2]) - float(a[2]))

    @staticmethod
    def transformQuat(out, a, q):
        """gl-matrix 3.4.1 vec3.transformQuat"""
        qx, qy, qz, qw = (float(v) for v in q)
        x, y, z = float(a[0]), float(a[1]), float(a[2])
        uvx, uvy, uvz = qy * z - qz * y, qz * x - qx * z, qx * y - qy * x
        uuvx, uuvy, uuvz = qy * uvz - qz * uvy, qz * uvx - qx * uvz, qx * uvy - qy * uvx
        w2 = qw * 2
        uvx *= w2; uvy *= w2; uvz *= w2
        uuvx *= 2; uuvy *= 2; uuvz *= 2
        out[0], out[1], out[2] = x + uvx + uuvx, y + uvy + uuvy, z + uvz + uuvz
        return out

    @staticmethod
    def cross(out, a, b):
        ax, ay, az = float(a[0]), float(a[1]), float(a[2])
        bx, by, bz = float(b[0]), float(b[1]), float(b[2])
        out[0], out[1], out[2] = ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx
        return out

    @staticmethod
    def dot(a, b):
        return float(a[0]) * float(b[0]) + float(a[1]) * float(b[1]) + float(a[2]) * float(b[2])

    @staticmethod
    def normalize(out, a):
        """gl-matrix 3.4.1 vec3.normalize"""
        x, y, z = float(a[0]), float(a[1]), float(a[2])
        length = x * x + y * y + z * z
        if length > 0:
            length = 1 / math.sqrt(length)
        out[0], out[1], out[2] = x * length, y * length, z * length
        return out


class quat:
    @staticmethod
    def create():
        return _f32([0, 0, 0, 1])

    @staticmethod
    def clone(a):
        return _f32([float(a[0]), float(a[1]), float(a[2]), float(a[3])])

    @staticmethod
    def invert(out, a):
        a0, a1, a2, a3 = (float(v) for v in a)
        dot = a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3
        inv = 1.0 / dot if dot else 0.0
        out[0], out[1], out[2], out[3] = -a0 * inv, -a1 * inv, -a2 * inv, a3 * inv
        return out

    @staticmethod
    def setAxisAngle(out, axis, rad):
        rad = rad * 0.5
        s = math.sin(rad)
        out[0], out[1], out[2], out[3] = s * axis[0], s * axis[1], s * axis[2], math.cos(rad)
        return out

    @staticmethod
    def rotateX(out, a, rad):
        """gl-matrix 3.4.1 quat.rotateX"""
        rad *= 0.5
        ax, ay, az, aw = (float(v) for v in a)
        bx, bw = math.sin(rad), math.cos(rad)
        out[0], out[1], out[2], out[3] = ax * bw + aw * bx, ay * bw + az * bx, az * bw - ay * bx, aw * bw - ax * bx
        return out

    @staticmethod
    def rotateY(out, a, rad):
        """gl-matrix 3.4.1 quat.rotateY"""
        rad *= 0.5
        ax, ay, az, aw = (float(v) for v in a)
        by, bw = math.sin(rad), math.cos(rad)
        out[0], out[1], out[2], out[3] = ax * bw - az * by, ay * bw + aw * by, az * bw + ax * by, aw * bw - ay * by
        return out

    @staticmethod
    def multiply(out, a, b):
        ax, ay, az, aw = (float(v) for v in a)
        bx, by, bz, bw = (float(v) for v in b)
        out[0] = ax * bw + aw * bx + ay * bz - az * by
        out[1] = ay * bw + aw * by + az * bx - ax * bz
        out[2] = az * bw + aw * bz + ax * by - ay * bx
        out[3] = aw * bw - ax * bx - ay * by - az * bz
        return out


class mat4:
    @staticmethod
    def create():
        out = np.zeros(16, dtype=np.float32)
        out[0] = out[5] = out[10] = out[15] = 1
        return out

    @staticmethod
    def fromTranslation(out, v):
        out[:] = 0
        out[0] = out[5] = out[10] = out[15] = 1
        out[12], out[13], out[14] = v[0], v[1], v[2]
        return out

    @staticmethod
    def fromScaling(out, v):
        out[:] = 0
        out[0], out[5], out[10], out[15] = v[0], v[1], v[2], 1
        return out

    @staticmethod
    def fromRotation(out, rad, axis):
        """gl-matrix 3.4.1 mat4.fromRotation (axis normalised with hypot; None for a degenerate axis)"""
        x, y, z = float(axis[0]), float(axis[1]), float(axis[2])
        length = math.hypot(x, y, z)
        if length < 0.000001:
            return None
        length = 1 / length
        x *= length; y *= length; z *= length
        s, c = math.sin(rad), math.cos(rad)
        t = 1 - c
        out[0] = x * x * t + c; out[1] = y * x * t + z * s; out[2] = z * x * t - y * s; out[3] = 0
        out[4] = x * y * t - z * s; out[5] = y * y * t + c; out[6] = z * y * t + x * s; out[7] = 0
        out[8] = x * z * t + y * s; out[9] = y * z * t - x * s; out[10] = z * z * t + c; out[11] = 0
        out[12] = out[13] = out[14] = 0; out[15] = 1
        return out

    @staticmethod
    def multiply(out, a, b):
        a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33 = (float(v) for v in a)
        bb = [float(v) for v in b]
        res = [0.0] * 16
        for c in range(4):
            b0, b1, b2, b3 = bb[4 * c:4 * c + 4]
            res[4 * c + 0] = b0 * a00 + b1 * a10 + b2 * a20 + b3 * a30
            res[4 * c + 1] = b0 * a01 + b1 * a11 + b2 * a21 + b3 * a31
            res[4 * c + 2] = b0 * a02 + b1 * a12 + b2 * a22 + b3 * a32
            res[4 * c + 3] = b0 * a03 + b1 * a13 + b2 * a23 + b3 * a33
        out[:] = res
        return out

    @staticmethod
    def invert(out, a):
        a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33 = (float(v) for v in a)
        b00 = a00 * a11 - a01 * a10
        b01 = a00 * a12 - a02 * a10
        b02 = a00 * a13 - a03 * a10
        b03 = a01 * a12 - a02 * a11
        b04 = a01 * a13 - a03 * a11
        b05 = a02 * a13 - a03 * a12
        b06 = a20 * a31 - a21 * a30
        b07 = a20 * a32 - a22 * a30
        b08 = a20 * a33 - a23 * a30
        b09 = a21 * a32 - a22 * a31
        b10 = a21 * a33 - a23 * a31
        b11 = a22 * a33 - a23 * a32
        det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06
        if not det:
            return None
        det = 1.0 / det
        res = [
            (a11 * b11 - a12 * b10 + a13 * b09) * det,
            (a02 * b10 - a01 * b11 - a03 * b09) * det,
            (a31 * b05 - a32 * b04 + a33 * b03) * det,
            (a22 * b04 - a21 * b05 - a23 * b03) * det,
            (a12 * b08 - a10 * b11 - a13 * b07) * det,
            (a00 * b11 - a02 * b08 + a03 * b07) * det,
            (a32 * b02 - a30 * b05 - a33 * b01) * det,
            (a20 * b05 - a22 * b02 + a23 * b01) * det,
            (a10 * b10 - a11 * b08 + a13 * b06) * det,
            (a01 * b08 - a00 * b10 - a03 * b06) * det,
            (a30 * b04 - a31 * b02 + a33 * b00) * det,
            (a21 * b02 - a20 * b04 - a23 * b00) * det,
            (a11 * b07 - a10 * b09 - a12 * b06) * det,
            (a00 * b09 - a01 * b07 + a02 * b06) * det,
            (a31 * b01 - a30 * b03 - a32 * b00) * det,
            (a20 * b03 - a21 * b01 + a22 * b00) * det,
        ]
        out[:] = res
        return out

    @staticmethod
    def perspective(out, fovy, aspect, near, far):
        f = 1.0 / math.tan(fovy / 2)
        out[:] = 0
        out[0] = f / aspect
        out[5] = f
        out[11] = -1
        if far is not None and far != math.inf:
            nf = 1 / (near - far)
            out[10] = (far + near) * nf
            out[14] = 2 * far * near * nf
        else:
            out[10] = -1
            out[14] = -2 * near
        return out

    @staticmethod
    def fromRotationTranslationScale(out, q, v, s):
        x, y, z, w = (float(c) for c in q)
        x2, y2, z2 = x + x, y + y, z + z
        xx, xy, xz = x * x2, x * y2, x * z2
        yy, yz, zz = y * y2, y * z2, z * z2
        wx, wy, wz = w * x2, w * y2, w * z2
        sx, sy, sz = float(s[0]), float(s[1]), float(s[2])
        out[0] = (1 - (yy + zz)) * sx
        out[1] = (xy + wz) * sx
        out[2] = (xz - wy) * sx
        out[3] = 0
        out[4] = (xy - wz) * sy
        out[5] = (1 - (xx + zz)) * sy
        out[6] = (yz + wx) * sy
        out[7] = 0
        out[8] = (xz + wy) * sz
        out[9] = (yz - wx) * sz
        out[10] = (1 - (xx + yy)) * sz
        out[11] = 0
        out[12], out[13], out[14] = v[0], v[1], v[2]
        out[15] = 1
        return out

    @staticmethod
    def scale(out, a, v):
        x, y, z = float(v[0]), float(v[1]), float(v[2])
        aa = [float(c) for c in a]
        res = [aa[0] * x, aa[1] * x, aa[2] * x, aa[3] * x,
               aa[4] * y, aa[5] * y, aa[6] * y, aa[7] * y,
               aa[8] * z, aa[9] * z, aa[10] * z, aa[11] * z,
               aa[12], aa[13], aa[14], aa[15]]
        out[:] = res
        return out

    @staticmethod
    def fromQuat(out, q):
        x, y, z, w = (float(c) for c in q)
        x2, y2, z2 = x + x, y + y, z + z
        xx, yx, yy = x * x2, y * x2, y * y2
        zx, zy, zz = z * x2, z * y2, z * z2
        wx, wy, wz = w * x2, w * y2, w * z2
        out[0] = 1 - yy - zz
        out[1] = yx + wz
        out[2] = zx - wy
        out[3] = 0
        out[4] = yx - wz
        out[5] = 1 - xx - zz
        out[6] = zy + wx
        out[7] = 0
        out[8] = zx + wy
        out[9] = zy - wx
        out[10] = 1 - xx - yy
        out[11] = 0
        out[12] = out[13] = out[14] = 0
        out[15] = 1
        return out

    @staticmethod
    def translate(out, a, v):
        x, y, z = float(v[0]), float(v[1]), float(v[2])
        aa = [float(c) for c in a]
        res = list(aa)
        res[12] = aa[0] * x + aa[4] * y + aa[8] * z + aa[12]
        res[13] = aa[1] * x + aa[5] * y + aa[9] * z + aa[13]
        res[14] = aa[2] * x + aa[6] * y + aa[10] * z + aa[14]
        res[15] = aa[3] * x + aa[7] * y + aa[11] * z + aa[15]
        out[:] = res
        return out


class Component(EventTarget):
    """src/js/Component.js"""

    def __init__(self, node=None):
        super().__init__()
        self.node = node


class Transform(Component):
    """src/js/Transform.js:4-116 (TRS with 'change' events).  As in the reference, ``node.parent`` is consulted
    for the global matrices; Node.js:9-11 never sets a usable parent on the transform's node handle, so the
    global matrices equal the local ones unless a caller builds a hierarchy explicitly."""

    def __init__(self, node=None):
        super().__init__(node)
        self.version = 0                      # bumped by every setter (lets callers cache derived matrices)
        self._localRotation = _f32([0, 0, 0, 1])
        self._localTranslation = _f32([0, 0, 0])
        self._localScale = _f32([1, 1, 1])

    @property
    def localRotation(self):
        return quat.clone(self._localRotation)

    @localRotation.setter
    def localRotation(self, value):
        self._localRotation = quat.clone(value)
        self.version += 1
        self.dispatchEvent(Event('change'))

    @property
    def localTranslation(self):
        return vec3.clone(self._localTranslation)

    @localTranslation.setter
    def localTranslation(self, value):
        self._localTranslation = vec3.clone(value)
        self.version += 1
        self.dispatchEvent(Event('change'))

    @property
    def localScale(self):
        return vec3.clone(self._localScale)

    @localScale.setter
    def localScale(self, value):
        self._localScale = vec3.clone(value)
        self.version += 1
        self.dispatchEvent(Event('change'))

    @property
    def globalTranslation(self):               # Transform.js:35-37 (mat4.getTranslation: elements 12..14)
        m = self.globalMatrix
        return _f32([float(m[12]), float(m[13]), float(m[14])])

    @property
    def localMatrix(self):
        return mat4.fromRotationTranslationScale(mat4.create(), self._localRotation, self._localTranslation, self._localScale)

    @property
    def globalMatrix(self):
        parent = getattr(self.node, 'parent', None)
        if parent is not None:
            g = parent.transform.globalMatrix
            return mat4.multiply(g, g, self.localMatrix)
        return self.localMatrix

    @property
    def inverseLocalMatrix(self):
        m = mat4.create()
        mat4.scale(m, m, vec3.inverse(vec3.create(), self._localScale))
        mat4.multiply(m, m, mat4.fromQuat(mat4.create(), quat.invert(quat.create(), self._localRotation)))
        mat4.translate(m, m, vec3.negate(vec3.create(), self._localTranslation))
        return m

    @property
    def inverseGlobalMatrix(self):
        parent = getattr(self.node, 'parent', None)
        if parent is not None:
            inv = parent.transform.inverseGlobalMatrix
            return mat4.multiply(inv, self.inverseLocalMatrix, inv)
        return self.inverseLocalMatrix


class Node:
    """src/js/Node.js:3-51"""

    def __init__(self):
        self.children = []
        self.parent = None
        self.components = [Transform(self)]

    def addChild(self, node):
        if node.parent:
            node.parent.removeChild(node)
        self.children.append(node)
        node.parent = self

    def removeChild(self, node):
        if node in self.children:
            self.children.remove(node)
            node.parent = None

    def getComponent(self, type_):
        for c in self.components:
            if isinstance(c, type_):
                return c
        return None

    @property
    def transform(self):
        return self.getComponent(Transform)


class PerspectiveCamera(Component):
    """src/js/PerspectiveCamera.js:4-19"""

    def __init__(self, node=None, options=None):
        super().__init__(node)
        options = options or {}
        self.fovy = options.get('fovy', 1)
        self.aspect = options.get('aspect', 1)
        self.near = options.get('near', 0.1)
        self.far = options.get('far', 100)

    @property
    def projectionMatrix(self):
        return mat4.perspective(mat4.create(), self.fovy, self.aspect, self.near, self.far)


def default_camera(aspect=1.0):
    """RenderingContext.js:38-40,121: camera node at (0,0,2) with a PerspectiveCamera, aspect = width/height."""
    node = Node()
    node.transform.localTranslation = [0, 0, 2]
    cam = PerspectiveCamera(node)
    cam.aspect = aspect
    node.components.append(cam)
    return node


def iso_light_direction(camera, volume_transform, light):
    """ISORenderer.js:152-166: the light direction is given in view space; take it to model space and normalise."""
    centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5])
    modelMatrix = volume_transform.globalMatrix
    viewMatrix = camera.transform.inverseGlobalMatrix
    matrix = mat4.create()
    mat4.multiply(matrix, centerMatrix, matrix)
    mat4.multiply(matrix, modelMatrix, matrix)
    mat4.multiply(matrix, viewMatrix, matrix)
    mat4.invert(matrix, matrix)
    out = vec3.transformMat4(vec3.create(), light, matrix)
    vec3.normalize(out, out)
    return out


def mvp_inverse_matrix(camera, volume_transform):
    """MIPRenderer.js:86-97 (= EAMRenderer.js:105-116, MCSRenderer.js:93-104, MCMRenderer.js:95-106,164-175)."""
    centerMatrix = mat4.fromTranslation(mat4.create(), [-0.5, -0.5, -0.5])
    modelMatrix = volume_transform.globalMatrix
    viewMatrix = camera.transform.inverseGlobalMatrix
    projectionMatrix = camera.getComponent(PerspectiveCamera).projectionMatrix
    matrix = mat4.create()
    mat4.multiply(matrix, centerMatrix, matrix)
    mat4.multiply(matrix, modelMatrix, matrix)
    mat4.multiply(matrix, viewMatrix, matrix)
    mat4.multiply(matrix, projectionMatrix, matrix)
    mat4.invert(matrix, matrix)
    return matrix
