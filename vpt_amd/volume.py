"""Volume — src/js/Volume.js:3-127 re-hosted on HIP device memory; driven by a reader (vpt_amd/readers.py) the way
RenderingContext.setVolume does (RenderingContext.js:124-134)."""
import ctypes as C

import numpy as np

from . import _native as N
from .property_bag import EventTarget, CustomEvent

from .readers import RAWReader, GL_RED, GL_R8, GL_RG, GL_RG8, GL_UNSIGNED_BYTE      # noqa: F401  (re-exported: the in-memory RAW form lives in readers.py)


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
        if modality['type'] != GL_UNSIGNED_BYTE or modality['format'] not in (GL_RED, GL_RG):
            raise RuntimeError('Unknown volume datatype: %s' % modality['type'])    # Volume.js:103
        fmt, channels = (N.FORMAT_RG8, 2) if modality['format'] == GL_RG else (N.FORMAT_R8, 1)
        h = C.c_void_p()
        N.check(L.vpt_volume_create(self._gl._h, dims['width'], dims['height'], dims['depth'], fmt, C.byref(h)))
        self.texture = h
        placements = modality['placements']
        for placement in placements:
            index, position = placement['index'], placement['position']
            data = np.ascontiguousarray(self._reader.readBlock(index), dtype=np.uint8)
            bd = self.metadata['blocks'][index]['dimensions']
            N.check(L.vpt_volume_upload_block(self.texture, position['x'], position['y'], position['z'],
                                              bd['width'], bd['height'], bd['depth'],
                                              data.ctypes.data_as(C.c_void_p), data.size))
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
        """Upload a [depth][height][width] (R8) or [depth][height][width][2] (RG8) uint8 array (host -> HBM once)."""
        array = np.ascontiguousarray(array, dtype=np.uint8)
        if array.ndim == 4 and array.shape[3] != 2:
            raise ValueError('a two-channel volume is [depth][height][width][2]')
        d, h, w = array.shape[:3]
        channels = 2 if array.ndim == 4 else 1
        vol = cls(gl, RAWReader(array, {'width': w * channels, 'height': h, 'depth': d}))      # slices of w * channels bytes
        L = N.lib()
        hnd = C.c_void_p()
        N.check(L.vpt_volume_create(gl._h, w, h, d, N.FORMAT_RG8 if channels == 2 else N.FORMAT_R8, C.byref(hnd)))
        vol.texture = hnd
        # chunk along z so one call stays < 2 GiB
        zs = max(1, (1 << 30) // (w * h * channels))
        for z0 in range(0, d, zs):
            z1 = min(d, z0 + zs)
            chunk = array[z0:z1]
            N.check(L.vpt_volume_upload_block(hnd, 0, 0, z0, w, h, z1 - z0, chunk.ctypes.data_as(C.c_void_p), chunk.size))
        N.check(L.vpt_volume_finalize(hnd))
        vol.metadata = vol._reader.readMetadata()
        vol.modality = vol.metadata['modalities'][0]
        if channels == 2:                                   # the slices were handed over as w * 2 bytes wide: restore the description
            vol.modality['dimensions']['width'] = w
            vol.modality['format'], vol.modality['internalFormat'] = GL_RG, GL_RG8
            for b in vol.metadata['blocks']:
                b['dimensions']['width'] = w
        vol.ready = True
        vol.setFilter(filter)
        return vol

    def set_wide_tables(self, wide):
        """force the > 4 GiB addressing variant of the kernels (automatic above 4 GiB of bricked data)"""
        N.check(N.lib().vpt_volume_set_wide_tables(self.texture, 1 if wide else 0))

    def bricked_bytes(self):
        n = C.c_uint64(0)
        N.check(N.lib().vpt_volume_bricked_bytes(self.texture, C.byref(n)))
        return n.value
