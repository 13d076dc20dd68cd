"""The transfer-function widget of the reference as DATA: a list of Gaussian "bumps" (the JSON its Save / Load buttons exchange,
src/js/ui/TransferFunction/TransferFunction.js:74-85) and the RGBA8 texture its canvas holds for `renderer.setTransferFunction`
(`render()` :110-121, src/glsl/TransferFunction.glsl:32-35).  No DOM, no handles, no dragging: what a user of the reference carries
over is the bump file; `texture()` turns it into the texels on the GPU (vpt_transfer_function_rasterize, include/vpt.h)."""
import ctypes as C
import json

import numpy as np

from . import _native as N


class TransferFunction:
    """bumps: [{'position': {'x', 'y'}, 'size': {'x', 'y'}, 'color': {'r', 'g', 'b', 'a'}}, ...] — the reference's own objects"""

    def __init__(self, gl, bumps=None, width=256, height=256):
        self._gl = gl
        self.transferFunctionWidth = int(width)          # TransferFunction.js:31-35
        self.transferFunctionHeight = int(height)
        self.bumps = []
        for b in (bumps or []):
            self.addBump(b)

    # ---- the widget's list operations (TransferFunction.js:127-176) ----
    def addBump(self, options=None):
        o = options or {}
        pos, size, col = o.get('position', {}), o.get('size', {}), o.get('color', {})
        bump = {'position': {'x': float(pos.get('x', 0.5)), 'y': float(pos.get('y', 0.5))},      # defaults: addBump() :129-144
                'size': {'x': float(size.get('x', 0.2)), 'y': float(size.get('y', 0.2))},
                'color': {'r': float(col.get('r', 1)), 'g': float(col.get('g', 0)), 'b': float(col.get('b', 0)), 'a': float(col.get('a', 1))}}
        self.bumps.append(bump)
        return len(self.bumps) - 1

    def removeBump(self, index):
        del self.bumps[index]

    def removeAllBumps(self):
        self.bumps = []

    def resizeTransferFunction(self, width, height):       # :101-108
        self.transferFunctionWidth, self.transferFunctionHeight = int(width), int(height)

    # ---- Save / Load (:74-85: the bump array as JSON, nothing else) ----
    def dumps(self):
        return json.dumps(self.bumps)

    def loads(self, text):
        data = json.loads(text)
        if not isinstance(data, list):
            raise ValueError('a transfer-function file is a JSON array of bumps')
        self.bumps = []
        for b in data:
            self.addBump(b)
        return self

    def save(self, path):
        with open(path, 'w') as f:
            f.write(self.dumps())

    def load(self, path):
        with open(path) as f:
            return self.loads(f.read())

    # ---- the canvas ----
    def packed(self):
        """[count][8] float32: position.xy, size.xy, color.rgba (struct vpt_tf_bump)"""
        a = np.zeros((len(self.bumps), 8), dtype=np.float32)
        for k, b in enumerate(self.bumps):
            a[k] = (b['position']['x'], b['position']['y'], b['size']['x'], b['size']['y'],
                    b['color']['r'], b['color']['g'], b['color']['b'], b['color']['a'])
        return a

    def texture(self, unpremultiply=True):
        """[height][width][4] uint8 for renderer.setTransferFunction: row 0 = the canvas's top row (position.y = 1), as texImage2D(canvas)
        transfers it; unpremultiply: the colour divided by alpha again, as a browser hands a premultiplied WebGL canvas over"""
        w, h = self.transferFunctionWidth, self.transferFunctionHeight
        bumps = self.packed()
        out = np.empty((h, w, 4), dtype=np.uint8)
        N.check(N.lib().vpt_transfer_function_rasterize(self._gl._h, bumps.ctypes.data_as(C.c_void_p), len(self.bumps), w, h,
                                                        1 if unpremultiply else 0, out.ctypes.data_as(C.c_void_p)))
        return out

    @property
    def value(self):                                       # `get value()` :123-125: what the application passes to setTransferFunction
        return self.texture()
