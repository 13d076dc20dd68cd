"""Volume — src/js/Volume.js:3-127 re-hosted on HIP device memory; driven by a reader (vpt_amd/readers.py) the way
RenderingContext.setVolume does (RenderingContext.js:124-134)."""
import ctypes as C

import numpy as np

from . import _native as N
from .property_bag import EventTarget, CustomEvent

from .readers import (RAWReader, GL_RED, GL_R8, GL_RG, GL_RG8, GL_UNSIGNED_BYTE, GL_RGB, GL_RGB8, GL_RGBA, GL_RGBA8,      # noqa: F401  (re-exported)
                      GL_FLOAT, GL_HALF_FLOAT, GL_R32F, GL_R16F)


def device_format(modality):
    """(native format, channels in the file, numpy dtype of a block) for a manifest's (format, type) — Volume.js:58-60 allocates whatever
    internalFormat the manifest names and :84-105 `_typize` maps the GL type to a typed array.  What a WebGL2 sampler3D can
    filter is what is taken here: UNSIGNED_BYTE and FLOAT / HALF_FLOAT (half widens to float exactly) with 1-4 channels — the
    shaders read .rg, so channels past the second are dropped on upload (R8, RG8; R32F, RG32F).  Integer and 16-bit normalised
    types cannot be sampled through a float sampler in WebGL2 and raise the reference's error."""
    t, f = modality['type'], modality['format']
    if t == GL_UNSIGNED_BYTE and f in (GL_RED, GL_RG, GL_RGB, GL_RGBA):
        n = {GL_RED: 1, GL_RG: 2, GL_RGB: 3, GL_RGBA: 4}[f]
        return (N.FORMAT_R8 if n == 1 else N.FORMAT_RG8), n, np.uint8
    if t in (GL_FLOAT, GL_HALF_FLOAT) and f in (GL_RED, GL_RG, GL_RGB, GL_RGBA):
        n = {GL_RED: 1, GL_RG: 2, GL_RGB: 3, GL_RGBA: 4}[f]
        return (N.FORMAT_R32F if n == 1 else N.FORMAT_RG32F), n, (np.float32 if t == GL_FLOAT else np.float16)
    raise RuntimeError('Unknown volume datatype: %s' % t)                   # Volume.js:103


class Volume(EventTarget):
    """Volume.js:3-127.  ``gl`` is a vpt_amd.Context.  ``getTexture()`` returns the native volume handle
    once ``ready`` (the reference returns the WebGLTexture), else None."""

    def __init__(self, gl, reader=None, options=None):
        super().__init__()
        self._gl = gl
        self._reader = reader
        self.metadata = None
        self.ready = False
        self.texture = None
        self.modality = None

    def destroy(self):
        if self.texture:
            N.lib().vpt_volume_destroy(self.texture)
            self.texture = None
            self.ready = False

    def readMetadata(self):
        if not self.metadata:
            self.metadata = self._reader.readMetadata()
        return self.metadata

    def readModality(self, modalityName):
        L = N.lib()
        self.ready = False
        if not self.metadata:
            self.readMetadata()
        modality = next((m for m in self.metadata['modalities'] if m['name'] == modalityName), None)
        if modality is None:
            raise RuntimeError("Modality '%s' does not exist" % modalityName)      # Volume.js:40
        self.modality = modality
        if self.texture:
            L.vpt_volume_destroy(self.texture)
            self.texture = None
        dims = modality['dimensions']
        fmt, nch, dtype = device_format(modality)
        h = C.c_void_p()
        N.check(L.vpt_volume_create(self._gl._h, dims['width'], dims['height'], dims['depth'], fmt, C.byref(h)))
        self.texture = h
        placements = modality['placements']
        for placement in placements:
            index, position = placement['index'], placement['position']
            raw = self._reader.readBlock(index)
            bd = self.metadata['blocks'][index]['dimensions']
            data = np.frombuffer(raw, dtype=dtype) if isinstance(raw, (bytes, bytearray, memoryview)) else np.asarray(raw).view(dtype).reshape(-1)
            if nch > 2:                                   # RGB8 / RGBA8: texture(uVolume, p).rg reads the first two channels
                data = data.reshape(-1, nch)[:, :2]
            if dtype is not np.uint8:
                data = data.astype(np.float32)            # HALF_FLOAT widens exactly
            data = np.ascontiguousarray(data)
            N.check(L.vpt_volume_upload_block(self.texture, position['x'], position['y'], position['z'],
                                              bd['width'], bd['height'], bd['depth'],
                                              data.ctypes.data_as(C.c_void_p), data.nbytes))
            progress = (index + 1) / len(placements)
            self.dispatchEvent(CustomEvent('progress', {'detail': progress}))
        N.check(L.vpt_volume_finalize(self.texture))
        self.ready = True

    def load(self):
        self.readModality('default')

    def getTexture(self):
        return self.texture if self.ready else None

    def setFilter(self, filter):
        if not self.texture:
            return
        N.check(N.lib().vpt_volume_set_filter(self.texture, N.FILTER_LINEAR if filter == 'linear' else N.FILTER_NEAREST))

    # ---- extension: whole-array upload (one block) for synthetic volumes ----
    @classmethod
    def from_array(cls, gl, array, filter='linear'):
        """Upload a [depth][height][width] (uint8: R8; float16 / float32: R32F) or [depth][height][width][2] (RG8 / RG32F) array (host -> HBM once)."""
        array = np.asarray(array)
        f32 = array.dtype.kind == 'f'
        array = np.ascontiguousarray(array, dtype=np.float32 if f32 else np.uint8)
        if array.ndim == 4 and array.shape[3] != 2:
            raise ValueError('a two-channel volume is [depth][height][width][2]')
        d, h, w = array.shape[:3]
        channels = 2 if array.ndim == 4 else 1
        vox = channels * (4 if f32 else 1)
        vol = cls(gl, RAWReader(array.view(np.uint8), {'width': w * vox, 'height': h, 'depth': d}))      # slices of w * vox bytes
        L = N.lib()
        hnd = C.c_void_p()
        fmt = (N.FORMAT_RG32F if channels == 2 else N.FORMAT_R32F) if f32 else (N.FORMAT_RG8 if channels == 2 else N.FORMAT_R8)
        N.check(L.vpt_volume_create(gl._h, w, h, d, fmt, C.byref(hnd)))
        vol.texture = hnd
        # chunk along z so one call stays < 2 GiB
        zs = max(1, (1 << 30) // (w * h * vox))
        for z0 in range(0, d, zs):
            z1 = min(d, z0 + zs)
            chunk = array[z0:z1]
            N.check(L.vpt_volume_upload_block(hnd, 0, 0, z0, w, h, z1 - z0, chunk.ctypes.data_as(C.c_void_p), chunk.nbytes))
        N.check(L.vpt_volume_finalize(hnd))
        vol.metadata = vol._reader.readMetadata()
        vol.modality = vol.metadata['modalities'][0]
        if vox > 1:                                         # the slices were handed over as w * vox bytes wide: restore the description
            vol.modality['dimensions']['width'] = w
            if f32 and channels == 2:
                vol.modality['format'], vol.modality['internalFormat'], vol.modality['type'] = GL_RG, 0x8230, GL_FLOAT      # RG32F
            elif f32:
                vol.modality['format'], vol.modality['internalFormat'], vol.modality['type'] = GL_RED, GL_R32F, GL_FLOAT
            else:
                vol.modality['format'], vol.modality['internalFormat'] = GL_RG, GL_RG8
            for b in vol.metadata['blocks']:
                b['dimensions']['width'] = w
        vol.ready = True
        vol.setFilter(filter)
        return vol

    def upload_block(self, x, y, z, block):
        """(extension) texSubImage3D of one more block into the ready volume: `block` is [depth][height][width] in the volume's texel type;
        the device layouts are rebuilt by the next pass that samples the volume"""
        block = np.ascontiguousarray(block)
        d, h, w = block.shape[:3]
        N.check(N.lib().vpt_volume_upload_block(self.texture, int(x), int(y), int(z), w, h, d, block.ctypes.data_as(C.c_void_p), block.nbytes))

    def set_wide_tables(self, wide):
        """force the > 4 GiB addressing variant of the kernels (automatic above 4 GiB of bricked data)"""
        N.check(N.lib().vpt_volume_set_wide_tables(self.texture, 1 if wide else 0))

    def bricked_bytes(self):
        n = C.c_uint64(0)
        N.check(N.lib().vpt_volume_bricked_bytes(self.texture, C.byref(n)))
        return n.value
