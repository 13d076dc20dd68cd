"""Tone mappers — host-side mirror of src/js/tonemappers/*.js over the C-ABI (vpt_tonemapper_*).

    ToneMapperFactory('artistic')(gl, texture, {'resolution': 512})

``gl`` is the vpt ``Context`` (the slot where the reference passes its WebGL2RenderingContext).  ``texture`` is what the
reference gets from ``renderer.getTexture()``: here either the renderer itself (its RGBA16F render buffer stays in HBM
and is read in place), a ``[rows][width][4]`` float16 array (uploaded), or None (the 1x1 white placeholder of
RenderingContext.js:176-181).  ``render()`` is one native pass; ``getTexture()`` returns the RGBA8 image read back.
Property names, labels, types, defaults and bounds are the reference's.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .property_bag import PropertyBag


def _f32(x):
    return float(np.float32(x))


class AbstractToneMapper(PropertyBag):
    """src/js/tonemappers/AbstractToneMapper.js:10-81"""
    _KIND = None

    def __init__(self, gl, texture, options=None):
        super().__init__()
        options = options or {}
        self._resolution = options['resolution'] if options.get('resolution') is not None else 512    # :15
        self._gl = gl
        self._h = None
        self._rebuildBuffers()
        self.setTexture(texture)

    def _size(self):
        r = self._resolution
        return (int(r), int(r)) if isinstance(r, (int, float)) else (int(r[0]), int(r[1]))

    def destroy(self):                                             # :28-33
        if self._h:
            N.lib().vpt_tonemapper_destroy(self._h)
            self._h = None

    def render(self):                                              # :35-38
        self._renderFrame()

    def setTexture(self, texture):                                 # :40-42
        L = N.lib()
        self._texture = texture
        if texture is None:
            N.check(L.vpt_tonemapper_set_source(self._h, None))
        elif hasattr(texture, '_h') and hasattr(texture, 'render'):      # a renderer: read its render buffer in place
            N.check(L.vpt_tonemapper_set_source(self._h, texture._h))
        else:
            img = np.ascontiguousarray(texture, dtype=np.float16)
            if img.ndim != 3 or img.shape[2] != 4:
                raise ValueError('texture must be a renderer or a [rows][width][4] float16 image')
            N.check(L.vpt_tonemapper_set_source_image(self._h, img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0]))

    def getTexture(self):                                          # :44-46 — the RGBA8 colour attachment, read back
        rows = C.c_int(0)
        N.check(N.lib().vpt_tonemapper_rows(self._h, C.byref(rows)))
        w, _ = self._size()
        out = np.empty((rows.value, w, 4), dtype=np.uint8)
        N.check(N.lib().vpt_tonemapper_read(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def _rebuildBuffers(self):                                     # :48-54
        w, h = self._size()
        if self._h is None:
            h_ = C.c_void_p()
            N.check(N.lib().vpt_tonemapper_create(self._gl._h, self._KIND, w, h, C.byref(h_)))
            self._h = h_
        else:
            N.check(N.lib().vpt_tonemapper_resize(self._h, w, h))

    def setResolution(self, resolution):                           # :56-61
        if resolution != self._resolution:
            self._resolution = resolution
            self._rebuildBuffers()

    def set_option(self, option, value):
        """extension: N.TONEMAPPER_OPTION_TABLE -> N.TONEMAPPER_TABLE_NEVER / _ALWAYS / _AUTO"""
        N.check(N.lib().vpt_tonemapper_set_option(self._h, int(option), int(value)))

    def _params(self):
        return N.TonemapParams(low=0.0, mid=0.5, high=1.0, saturation=1.0, min=0.0, max=1.0, exposure=1.0, gamma=2.2)

    def _renderFrame(self):                                        # :63-65, implemented by the subclasses' uniforms
        N.check(N.lib().vpt_tonemapper_render(self._h, C.byref(self._params())))


class ArtisticToneMapper(AbstractToneMapper):
    """src/js/tonemappers/ArtisticToneMapper.js:10-85"""
    _KIND = N.TONEMAPPER_ARTISTIC

    def __init__(self, gl, texture, options=None):
        super().__init__(gl, texture, options)
        self.registerProperties([                                  # :15-49
            {'name': 'low', 'label': 'Low', 'type': 'spinner', 'value': 0},
            {'name': 'high', 'label': 'High', 'type': 'spinner', 'value': 1},
            {'name': 'mid', 'label': 'Midtones', 'type': 'slider', 'value': 0.5, 'min': 0.00001, 'max': 0.99999},
            {'name': 'saturation', 'label': 'Saturation', 'type': 'spinner', 'value': 1},
            {'name': 'gamma', 'label': 'Gamma', 'type': 'spinner', 'value': 2.2, 'min': 0},
        ])

    def _params(self):                                             # :75-79
        p = super()._params()
        p.low, p.mid, p.high = _f32(self.low), _f32(self.mid), _f32(self.high)
        p.saturation, p.gamma = _f32(self.saturation), _f32(self.gamma)
        return p


class RangeToneMapper(AbstractToneMapper):
    """src/js/tonemappers/RangeToneMapper.js:10-66"""
    _KIND = N.TONEMAPPER_RANGE

    def __init__(self, gl, texture, options=None):
        super().__init__(gl, texture, options)
        self.registerProperties([                                  # :14-34
            {'name': 'min', 'label': 'Min', 'type': 'spinner', 'value': 0},
            {'name': 'max', 'label': 'Max', 'type': 'spinner', 'value': 1},
            {'name': 'gamma', 'label': 'Gamma', 'type': 'spinner', 'value': 2.2, 'min': 0},
        ])

    def _params(self):                                             # :58-60
        p = super()._params()
        p.min, p.max, p.gamma = _f32(self.min), _f32(self.max), _f32(self.gamma)
        return p


class _ExposureGammaToneMapper(AbstractToneMapper):
    """the eight curve mappers share one host class body (e.g. src/js/tonemappers/ReinhardToneMapper.js:10-59)"""

    def __init__(self, gl, texture, options=None):
        super().__init__(gl, texture, options)
        self.registerProperties([                                  # :14-29
            {'name': 'exposure', 'label': 'Exposure', 'type': 'spinner', 'value': 1, 'min': 0},
            {'name': 'gamma', 'label': 'Gamma', 'type': 'spinner', 'value': 2.2, 'min': 0},
        ])

    def _params(self):                                             # :53-54
        p = super()._params()
        p.exposure, p.gamma = _f32(self.exposure), _f32(self.gamma)
        return p


class ReinhardToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_REINHARD


class Reinhard2ToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_REINHARD2


class Uncharted2ToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_UNCHARTED2


class FilmicToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_FILMIC


class UnrealToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_UNREAL


class AcesToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_ACES


class LottesToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_LOTTES


class UchimuraToneMapper(_ExposureGammaToneMapper):
    _KIND = N.TONEMAPPER_UCHIMURA


def ToneMapperFactory(which):
    """src/js/tonemappers/ToneMapperFactory.js:12-27"""
    table = {
        'artistic': ArtisticToneMapper, 'range': RangeToneMapper, 'reinhard': ReinhardToneMapper,
        'reinhard2': Reinhard2ToneMapper, 'uncharted2': Uncharted2ToneMapper, 'filmic': FilmicToneMapper,
        'unreal': UnrealToneMapper, 'aces': AcesToneMapper, 'lottes': LottesToneMapper, 'uchimura': UchimuraToneMapper,
    }
    if which not in table:
        raise RuntimeError('No suitable class')
    return table[which]
