"""Animators — src/js/animators/CircleAnimator.js: the time-driven camera animator the reference's animation recorder
steps (RenderingContext.js:283-284 `this.cameraAnimator.update(t)`).  Pinned bit for bit by
tests/golden/circle_animator_r01.json (the reference's own CircleAnimator run under node).
OrbitCameraAnimator is the reference's interactive camera; here it is headless: the same state,
handlers and arithmetic, fed with event-like dicts and an explicit clock instead of DOM events and Date.now() —
pinned bit for bit by tests/golden/orbit_animator_r01.json (the reference's class run under node on scripted input)."""
import math
import time

from .scene import mat4, quat, vec3


class CircleAnimator:
    """src/js/animators/CircleAnimator.js:3-42"""

    def __init__(self, node, options=None):
        self.node = node
        self.center = [0, 0, 0]
        self.direction = [1, 0, 0]
        self.radius = 1
        self.frequency = 1
        for k, v in (options or {}).items():
            setattr(self, k, v)

    def _tilt(self):
        """the rotation taking +z onto `direction`, the reference's way: the UN-normalised quaternion (z x d, z . d)"""
        z = [0, 0, 1]
        d = vec3.normalize(vec3.create(), self.direction)
        c = vec3.cross(vec3.create(), z, d)
        return [float(c[0]), float(c[1]), float(c[2]), vec3.dot(z, d)]

    def update(self, t):
        """a point on the circle (radius, center, plane normal `direction`), `frequency` turns per unit of t — through the same chain
        of float32 4x4 products as the reference's update() (:16-40: shift * tilt * spin * scale), which the fixture's last bits follow"""
        tilt = self._tilt()
        r = self.radius
        factors = [mat4.fromTranslation(mat4.create(), self.center),
                   mat4.fromQuat(mat4.create(), tilt),
                   mat4.fromRotation(mat4.create(), self.frequency * t * 2 * math.pi, [0, 0, 1]),
                   mat4.fromScaling(mat4.create(), [r, r, r])]
        pose = mat4.create()
        for f in factors:
            mat4.multiply(pose, pose, f)
        start = [1, 0, 0]                                                         # a plain array: no float32 rounding before the setter
        self.node.transform.localTranslation = vec3.transformMat4(start, start, pose)
        self.node.transform.localRotation = tilt


class OrbitCameraAnimator:
    """src/js/animators/OrbitCameraAnimator.js:4-200.  `domElement` is kept for signature parity and never touched (no DOM:
    the caller invokes the `_handle*` methods with dicts carrying the fields the handlers read); `options['now']` is the
    clock `_update()` reads in milliseconds (default: wall clock, as Date.now())."""

    def __init__(self, camera, domElement=None, options=None):
        self.rotationSpeed = 0.005                                                # :14-19
        self.translationSpeed = 0.005
        self.moveSpeed = 0.001
        self.zoomSpeed = 0.001
        self.now = lambda: time.time() * 1000.0
        for k, v in (options or {}).items():
            setattr(self, k, v)
        self._camera = camera
        self._domElement = domElement
        self._focus = [0, 0, 0]
        self._focusDistance = vec3.distance(self._focus, self._camera.transform.globalTranslation)   # :25
        self._yaw = 0
        self._pitch = 0
        self._forward = self._backward = self._left = self._right = False
        self._isTranslating = False
        self._isRotating = False
        self._time = self.now()

    def _handlePointerDown(self, e):                                              # :50-60
        if e.get('button') == 0:
            self._isRotating = True
        elif e.get('button') == 1:
            self._isTranslating = True

    def _handlePointerUp(self, e=None):                                           # :62-69
        self._isTranslating = False
        self._isRotating = False

    def _handlePointerMove(self, e):                                              # :71-95
        dx = e.get('movementX', 0)
        dy = e.get('movementY', 0)
        if self._isRotating:
            angleX = -dx * self.rotationSpeed
            angleY = -dy * self.rotationSpeed
            self._rotateAroundFocus(angleX, angleY)                               # both branches of the shiftKey test do this
        if self._isTranslating:
            # :88-92 multiplies a number by the focus ARRAY (NaN) and hands _move a number: vec3.transformQuat then throws
            # a TypeError in the reference (strict mode, property store on a primitive) before anything is modified
            raise TypeError("Cannot create property '0' on number 'NaN'")

    def _handleWheel(self, e):                                                    # :97-99
        self._zoom(e.get('deltaY', 0) * self.zoomSpeed)

    def _handleKeyDown(self, e):                                                  # :101-108
        self._key(e, True)

    def _handleKeyUp(self, e):                                                    # :110-117
        self._key(e, False)

    def _key(self, e, down):
        k = str(e.get('key', '')).lower()
        if k == 'w':
            self._forward = down
        elif k == 'a':
            self._left = down
        elif k == 's':
            self._backward = down
        elif k == 'd':
            self._right = down

    def _updateCamera(self):                                                      # :119-131
        transform = self._camera.transform
        rotation = quat.create()
        quat.rotateY(rotation, rotation, self._yaw)
        quat.rotateX(rotation, rotation, self._pitch)
        translation = vec3.transformQuat(vec3.create(), [0, 0, self._focusDistance], rotation)
        transform.localRotation = rotation
        transform.localTranslation = vec3.add(vec3.create(), self._focus, translation)

    def _rotateAroundFocus(self, dx, dy):                                         # :133-144
        twopi = math.pi * 2
        halfpi = math.pi / 2
        self._pitch += dy
        self._pitch = min(max(self._pitch, -halfpi), halfpi)
        self._yaw += dx
        self._yaw = math.fmod(math.fmod(self._yaw, twopi) + twopi, twopi)         # JS %: the sign of the dividend
        self._updateCamera()

    def _move(self, v):                                                           # :146-153
        rotation = quat.create()
        quat.rotateY(rotation, rotation, self._yaw)
        quat.rotateX(rotation, rotation, self._pitch)
        vec3.transformQuat(v, v, rotation)
        vec3.add(self._focus, self._focus, v)
        self._updateCamera()

    def _zoom(self, amount):                                                      # :155-158
        self._focusDistance *= math.exp(amount)
        self._updateCamera()

    def _update(self):                                                            # :160-186 (the Ticker callback)
        t = self.now()
        dt = t - self._time
        self._time = t
        dx = 0
        dz = 0
        if self._forward:
            dz -= self.moveSpeed * self._focusDistance * dt
        if self._backward:
            dz += self.moveSpeed * self._focusDistance * dt
        if self._left:
            dx -= self.moveSpeed * self._focusDistance * dt
        if self._right:
            dx += self.moveSpeed * self._focusDistance * dt
        if dx != 0 or dz != 0:
            self._move([dx, 0, dz])

    def update(self, t):                                                          # :188-190
        pass                                                                      # "does not animate, it only responds to user input"
