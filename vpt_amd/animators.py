"""Animators — src/js/animators/CircleAnimator.js: the time-driven camera animator the reference's animation recorder
steps (RenderingContext.js:283-284 `this.cameraAnimator.update(t)`).  Pinned bit for bit by
tests/golden/circle_animator_r01.json (the reference's own CircleAnimator run under node).
(OrbitCameraAnimator is pointer / keyboard / wall-clock driven — UI, not built.)"""
import math

from .scene import mat4, vec3


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

    def update(self, t):                                                          # :16-40
        scale = mat4.fromScaling(mat4.create(), [self.radius, self.radius, self.radius])
        angle = self.frequency * t * 2 * math.pi
        phase = mat4.fromRotation(mat4.create(), angle, [0, 0, 1])
        from_ = [0, 0, 1]
        to = vec3.normalize(vec3.create(), self.direction)
        axis = vec3.cross(vec3.create(), from_, to)
        slant = vec3.dot(from_, to)
        orientationQuat = [float(axis[0]), float(axis[1]), float(axis[2]), slant]
        orientation = mat4.fromQuat(mat4.create(), orientationQuat)
        translation = mat4.fromTranslation(mat4.create(), self.center)
        composite = mat4.create()
        mat4.multiply(composite, composite, translation)
        mat4.multiply(composite, composite, orientation)
        mat4.multiply(composite, composite, phase)
        mat4.multiply(composite, composite, scale)
        position = [1, 0, 0]                                                      # a plain array: no float32 rounding before the setter
        self.node.transform.localTranslation = vec3.transformMat4(position, position, composite)
        self.node.transform.localRotation = orientationQuat
