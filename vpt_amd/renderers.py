"""Renderer plugin surface — AbstractRenderer + MIP / EAM / MCS / MCM + RendererFactory.

Host-side mirror of src/js/renderers/{Abstract,MIP,EAM,MCS,MCM}Renderer.js and RendererFactory.js:
same constructor signature, methods, hook names, property names/defaults and 'change' behaviour.
Every hook is a thin call into the C-ABI (include/vpt.h); the WebGL framebuffers of the reference
(SingleBuffer.js / DoubleBuffer.js) are HIP device buffers owned by the native renderer.

Differences that are part of the contract:
  * ``gl`` is a vpt_amd.Context; ``environmentTexture`` is an RGBA8 array [h][w][4] or None (1x1 white,
    RenderingContext.js:90-101).
  * ``options['resolution']`` is an int (square, the reference's behaviour) or (width, height).
  * the values the reference draws with Math.random() come from ``self.rng`` (option 'rng', default random.random);
    install a seeded callable for reproducible runs.
  * ``render()`` launches one fused kernel (generate -> integrate -> renderFrame) unless a subclass
    overrides a hook or ``self.fused`` is False; the two forms give identical buffers.
"""
import ctypes as C
import math
import random as _random

import numpy as np

from . import _native as N
from .property_bag import PropertyBag
from .scene import Transform, mvp_inverse_matrix


def _f32(x):
    return float(np.float32(x))


class AbstractRenderer(PropertyBag):
    _KIND = None

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__()
        options = options or {}
        res = options.get('resolution', 512)                       # AbstractRenderer.js:20
        self._resolution = res
        self._gl = gl
        self._volume = volume
        self._camera = camera
        self._environmentTexture = environmentTexture
        self._volumeTransform = options.get('transform') or Transform()   # AbstractRenderer.js:27
        self.rng = options.get('rng', _random.random)
        self.fused = options.get('fused', True)
        self._shard = options.get('shard')                         # (rank, world, rows_per_block) or None
        self._h = None
        self._rebuildBuffers()
        if environmentTexture is not None:
            self._upload_environment(environmentTexture)
        self._bound_volume = None

    # ---- native plumbing ------------------------------------------------------------------
    def _size(self):
        r = self._resolution
        if isinstance(r, (tuple, list)):
            return int(r[0]), int(r[1])
        if isinstance(r, dict):
            return int(r['width']), int(r['height'])
        return int(r), int(r)

    def _rebuildBuffers(self):                                     # AbstractRenderer.js:78-92
        L = N.lib()
        w, h = self._size()
        if self._h is None:
            hnd = C.c_void_p()
            N.check(L.vpt_renderer_create(self._gl._h, self._KIND, w, h, C.byref(hnd)))
            self._h = hnd
            if self._shard:
                N.check(L.vpt_renderer_set_shard(self._h, *self._shard))
        else:
            N.check(L.vpt_renderer_resize(self._h, w, h))

    def _upload_environment(self, tex):
        tex = np.ascontiguousarray(tex, dtype=np.uint8)
        assert tex.ndim == 3 and tex.shape[2] == 4, 'environment texture is [h][w][4] RGBA8'
        N.check(N.lib().vpt_renderer_set_environment(self._h, tex.ctypes.data_as(C.c_void_p), tex.shape[1], tex.shape[0]))

    def setEnvironmentMap(self, image):
        """the reference re-fills the context-owned environment texture in place (RenderingContext.js:135-140); the
        renderer holds a device copy here, so the context hands the new image down"""
        self._environmentTexture = image
        self._upload_environment(image)

    def _bind_volume(self):
        tex = self._volume.getTexture() if self._volume is not None else None
        if tex is not self._bound_volume:
            N.check(N.lib().vpt_renderer_set_volume(self._h, tex))
            self._bound_volume = tex

    def _matrix(self):
        # the reference rebuilds the matrix every frame (MIPRenderer.js:86-97); the result only depends on these
        # inputs, so it is cached until one of them changes
        from .scene import PerspectiveCamera
        pc = self._camera.getComponent(PerspectiveCamera)
        key = (self._camera.transform.version, self._volumeTransform.version, id(self._camera), id(self._volumeTransform),
               pc.fovy, pc.aspect, pc.near, pc.far,
               getattr(self._camera, 'parent', None) is None, getattr(self._volumeTransform.node, 'parent', None) is None)
        if getattr(self, '_matrix_key', None) != key or not key[-1] or not key[-2]:
            self._matrix_cache = mvp_inverse_matrix(self._camera, self._volumeTransform)
            self._matrix_key = key
        return self._matrix_cache

    def _new_uniforms(self):
        u = N.Uniforms()
        m = self._matrix()
        C.memmove(u.mvp_inverse, m.ctypes.data, 64)
        return u

    def _hooks_overridden(self):
        # a user subclass that overrides a hook of the shipped class must see its hook called
        base = type(self)._BASE
        return any(getattr(type(self), n) is not getattr(base, n)
                   for n in ('_generateFrame', '_integrateFrame', '_renderFrame'))

    # ---- AbstractRenderer.js public surface ---------------------------------------------------
    def destroy(self):                                             # :51-58
        if self._h:
            N.lib().vpt_renderer_destroy(self._h)
            self._h = None

    def render(self):                                              # :60-70
        if self.fused and not self._hooks_overridden():
            self._renderFused()
            return
        self._generateFrame()
        self._integrateFrame()
        self._renderFrame()

    def reset(self):                                               # :72-76
        self._resetFrame()

    def setVolume(self, volume):                                   # :94-97
        self._volume = volume
        self.reset()

    def setTransferFunction(self, transferFunction):               # :99-104 (texImage2D SRGB8_ALPHA8)
        tf = np.ascontiguousarray(transferFunction, dtype=np.uint8)
        if tf.ndim != 3 or tf.shape[2] != 4:
            raise TypeError('transfer function is an RGBA8 image [h][w][4]')
        N.check(N.lib().vpt_renderer_set_transfer_function(self._h, tf.ctypes.data_as(C.c_void_p), tf.shape[1], tf.shape[0]))

    def setResolution(self, resolution):                           # :106-112
        if resolution != self._resolution:
            self._resolution = resolution
            self._rebuildBuffers()
            self.reset()

    def getTexture(self):                                          # :114-116 -> RGBA16F image [rows][w][4]
        return self.read(N.BUFFER_RENDER)

    # ---- hooks (IMPLEMENT in subclasses, AbstractRenderer.js:118-140) ---------------------------
    def _resetFrame(self): pass
    def _generateFrame(self): pass
    def _integrateFrame(self): pass
    def _renderFrame(self): pass
    def _renderFused(self): pass

    # ---- buffer specs (AbstractRenderer.js:134-155 and the subclasses' _get*BufferSpec) --------------------
    # In the reference these lists drive the allocation of the WebGL attachments (_rebuildBuffers, :78-92); here the native
    # renderer owns the HIP buffers and the hooks DESCRIBE them with the reference's GL enums: one dict per attachment, in
    # attachment order (what vpt_renderer_read returns for BUFFER_FRAME / BUFFER_ACCUM / BUFFER_MCM_* / BUFFER_RENDER).
    _GL = {'NEAREST': 9728, 'CLAMP_TO_EDGE': 33071, 'RED': 6403, 'RG': 33319, 'RGBA': 6408, 'R8': 33321, 'R32F': 33326, 'RG32F': 33328,
           'RGBA16F': 34842, 'RGBA32F': 34836, 'UNSIGNED_BYTE': 5121, 'FLOAT': 5126}
    _BUFFER_FORMATS = {          # kind -> (frame attachments, accumulation attachments) as (format, iformat, type)
        N.RENDERER_MIP: ([('RED', 'R8', 'UNSIGNED_BYTE')], [('RED', 'R8', 'UNSIGNED_BYTE')]),                     # MIPRenderer.js:133-157
        N.RENDERER_EAM: ([('RGBA', 'RGBA', 'UNSIGNED_BYTE')], [('RGBA', 'RGBA', 'UNSIGNED_BYTE')]),                # EAMRenderer.js:155-179
        N.RENDERER_LAO: ([('RGBA', 'RGBA', 'UNSIGNED_BYTE')], [('RGBA', 'RGBA', 'UNSIGNED_BYTE')]),                # LAORenderer.js
        N.RENDERER_MCS: ([('RGBA', 'RGBA32F', 'FLOAT')], [('RGBA', 'RGBA32F', 'FLOAT')]),                          # MCSRenderer.js:156-180
        N.RENDERER_MCM: ([('RGBA', 'RGBA32F', 'FLOAT')], [('RGBA', 'RGBA32F', 'FLOAT')] * 4),                      # MCMRenderer.js:201-263
        N.RENDERER_ISO: ([('RGBA', 'RGBA16F', 'FLOAT')], [('RGBA', 'RGBA16F', 'FLOAT')]),                          # ISORenderer.js
        N.RENDERER_DEPTH: ([('RED', 'R32F', 'FLOAT')], [('RED', 'R32F', 'FLOAT')]),                                # DepthRenderer.js
        N.RENDERER_DOS: ([('RGBA', 'RGBA32F', 'FLOAT')], [('RGBA', 'RGBA32F', 'FLOAT'), ('RED', 'R32F', 'FLOAT')]),   # DOSRenderer.js:277-305
    }

    def _spec(self, fmt):
        g = self._GL
        w, h = self._size()
        return {'width': w, 'height': h, 'min': g['NEAREST'], 'mag': g['NEAREST'],
                'format': g[fmt[0]], 'iformat': g[fmt[1]] if fmt[1] in g else g['RGBA'], 'type': g[fmt[2]]}

    def _getFrameBufferSpec(self):
        return [self._spec(f) for f in self._BUFFER_FORMATS[self._KIND][0]]

    def _getAccumulationBufferSpec(self):
        return [self._spec(f) for f in self._BUFFER_FORMATS[self._KIND][1]]

    def _getRenderBufferSpec(self):                                # AbstractRenderer.js:142-155
        g = self._GL
        d = self._spec(('RGBA', 'RGBA16F', 'FLOAT'))
        d['wrapS'] = d['wrapT'] = g['CLAMP_TO_EDGE']
        return [d]

    # ---- extensions: read-back and counters -----------------------------------------------------
    def local_rows(self):
        n = C.c_int(0)
        N.check(N.lib().vpt_renderer_local_rows(self._h, C.byref(n)))
        return n.value

    def global_rows(self):
        """global row index of every local row (-1 = padding row of a shard)"""
        out = []
        g = C.c_int(0)
        for l in range(self.local_rows()):
            N.check(N.lib().vpt_renderer_global_row(self._h, l, C.byref(g)))
            out.append(g.value)
        return np.array(out, dtype=np.int64)

    def read(self, buffer):
        w, _ = self._size()
        rows = self.local_rows()
        if buffer == N.BUFFER_RENDER:
            arr = np.empty((rows, w, 4), dtype=np.float16)
        elif buffer in (N.BUFFER_FRAME, N.BUFFER_ACCUM):
            if self._KIND == N.RENDERER_MIP:
                arr = np.empty((rows, w), dtype=np.uint8)
            elif self._KIND in (N.RENDERER_EAM, N.RENDERER_LAO):
                arr = np.empty((rows, w, 4), dtype=np.uint8)
            elif self._KIND == N.RENDERER_ISO:
                arr = np.empty((rows, w, 4), dtype=np.float16)
            elif self._KIND == N.RENDERER_DEPTH:
                arr = np.empty((rows, w), dtype=np.float32)
            elif self._KIND == N.RENDERER_DOS and buffer == N.BUFFER_FRAME:
                raise N.VptError(N.ERR_INVALID, "the DOS renderer's frame buffer is never written")
            else:
                arr = np.empty((rows, w, 4), dtype=np.float32)
        elif buffer == N.BUFFER_DOS_OCCLUSION:
            arr = np.empty((rows, w), dtype=np.float32)
        else:
            arr = np.empty((rows, w, 4), dtype=np.float32)
        N.check(N.lib().vpt_renderer_read(self._h, buffer, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return arr

    def sample_count(self):
        n = C.c_uint64(0)
        N.check(N.lib().vpt_renderer_sample_count(self._h, C.byref(n)))
        return n.value

    def clear_sample_count(self):
        N.check(N.lib().vpt_renderer_clear_sample_count(self._h))

    def tile_classes(self):
        """(HIT tiles, MISS tiles, violations) of the tile classes in force (vpt_renderer_tile_classes)"""
        h, m, v = C.c_int(0), C.c_int(0), C.c_uint64(0)
        N.check(N.lib().vpt_renderer_tile_classes(self._h, C.byref(h), C.byref(m), C.byref(v)))
        return h.value, m.value, v.value

    def set_profiling(self, enabled):
        """False/0: off; True/1: time every launch; n > 1: every n-th launch"""
        N.check(N.lib().vpt_renderer_set_profiling(self._h, int(enabled)))

    def profile(self):
        ms, n = C.c_double(0), C.c_uint32(0)
        N.check(N.lib().vpt_renderer_profile(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_side(self):
        """(total ms, launches) of the sampled passes' first side-stream launch (tile classes: the MISS-tile kernel)"""
        ms, n = C.c_double(0), C.c_uint32(0)
        N.check(N.lib().vpt_renderer_profile_side(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def render_buffer_device(self):
        p, n = C.c_void_p(), C.c_size_t(0)
        N.check(N.lib().vpt_renderer_render_buffer_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _collect_frames(self, count):
        """draws the per-frame uniforms of the next `count` frames exactly as `count` render() calls would"""
        vars_ = np.zeros((count, 8), dtype=np.float32)
        u = None
        for k in range(count):
            u = self._prepare_frame_uniforms()
            vars_[k, 0], vars_[k, 1], vars_[k, 2] = u.rand_seed, u.offset, u.mix
            vars_[k, 4:7] = list(u.light_direction)
        return u, vars_

    def play(self, count, use_graph=True, fused=False, frames=False):
        """`count` render() passes enqueued by one native call: eager launches, one hipGraph replay (use_graph: where the library
        knows the graph to be the faster form — one stream, no tile classes — else eager all the same), one launch
        running all passes with the photon state / accumulator in registers (fused), or — MCM — the same with EVERY pass's
        frame written to the renderer's frame ring (frames; read_frame_slot); same buffers as count x render()"""
        self._bind_volume()
        u, vars_ = self._collect_frames(count)
        mode = N.PLAY_FRAMES if frames else (N.PLAY_FUSED if fused else (N.PLAY_GRAPH if use_graph else N.PLAY_EAGER))
        N.check(N.lib().vpt_renderer_play(self._h, C.byref(u), vars_.ctypes.data_as(C.c_void_p), count, mode))

    def play_into(self, count, first_target, stride_bytes):
        """`count` eager render() passes by one native call, frame i into caller-owned device memory at first_target + i * stride_bytes"""
        self._bind_volume()
        u, vars_ = self._collect_frames(count)
        N.check(N.lib().vpt_renderer_play_into(self._h, C.byref(u), vars_.ctypes.data_as(C.c_void_p), count, C.c_void_p(first_target), stride_bytes))

    def play_into_display(self, tone_mapper, count, first_target, stride_bytes):
        """play_into with the frames as the armed `tone_mapper` shows them: RGBA8, frame i at first_target + i * stride_bytes (vpt_renderer_play_into_display)"""
        self._bind_volume()
        u, vars_ = self._collect_frames(count)
        N.check(N.lib().vpt_renderer_play_into_display(self._h, tone_mapper._h, C.byref(u), vars_.ctypes.data_as(C.c_void_p), count,
                                                       C.c_void_p(first_target), stride_bytes))

    def bucket_launches(self):
        """buckets of frames run by the bucket kernels so far (OPTION_BUCKET_KERNEL)"""
        n = C.c_uint64(0)
        N.check(N.lib().vpt_renderer_bucket_launches(self._h, C.byref(n)))
        return n.value

    def read_frame_slot(self, slot):
        """frame `slot` of the last play(frames=True) call: [local rows][W][4] float16"""
        out = np.empty((self.local_rows(), self._size()[0], 4), dtype=np.float16)
        N.check(N.lib().vpt_renderer_read_frame_slot(self._h, int(slot), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def set_option(self, option, value):
        N.check(N.lib().vpt_renderer_set_option(self._h, int(option), int(value)))

    def set_render_target(self, device_ptr, nbytes):
        """redirect _renderFrame output into caller-owned device memory (None restores the own buffer)"""
        N.check(N.lib().vpt_renderer_set_render_target(self._h, C.c_void_p(device_ptr) if device_ptr else None, nbytes))

    def join(self):
        """join the side streams of split passes into the context's stream (vpt_renderer_join)"""
        N.check(N.lib().vpt_renderer_join(self._h))

    def probe_sample(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.empty((xyz.shape[0], 4), dtype=np.float32)
        self._bind_volume()
        N.check(N.lib().vpt_probe_sample(self._h, xyz.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), xyz.shape[0]))
        return out

    def probe_sample_boundary(self, xyz):
        """the same samples with the positions outside the cube taken from the volume's boundary atlas (vpt_probe_sample_boundary)"""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.empty((xyz.shape[0], 4), dtype=np.float32)
        self._bind_volume()
        N.check(N.lib().vpt_probe_sample_boundary(self._h, xyz.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), xyz.shape[0]))
        return out


AbstractRenderer._BASE = AbstractRenderer

_TF_PROPERTY = {'name': 'transferFunction', 'label': 'Transfer function', 'type': 'transfer-function',
                'value': np.zeros(256, dtype=np.uint8)}


def _install_change_handler(renderer, reset_on):
    def on_change(e):
        name = e.detail['name']
        if name == 'transferFunction':
            renderer.setTransferFunction(renderer.transferFunction)
        if name in reset_on:
            renderer.reset()
    renderer.addEventListener('change', on_change)


class MIPRenderer(AbstractRenderer):
    """src/js/renderers/MIPRenderer.js:13-159"""
    _KIND = N.RENDERER_MIP

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([
            {'name': 'steps', 'label': 'Steps', 'type': 'spinner', 'value': 64, 'min': 1},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('transferFunction',))      # :34-46 (steps does NOT reset)

    def _resetFrame(self):                                         # :61-68
        N.check(N.lib().vpt_renderer_reset(self._h, None))

    def _prepare_generate(self):                                   # :82-97
        u = self._new_uniforms()
        u.step_size = _f32(1 / self.steps)
        u.offset = _f32(self.rng())
        self._u = u
        return u

    def _generateFrame(self):                                      # :69-100
        self._bind_volume()
        N.check(N.lib().vpt_renderer_generate(self._h, C.byref(self._prepare_generate())))

    def _integrateFrame(self):                                     # :102-117
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._u)))

    def _renderFrame(self):                                        # :119-131
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def _prepare_frame_uniforms(self):
        return self._prepare_generate()

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))


class EAMRenderer(AbstractRenderer):
    """src/js/renderers/EAMRenderer.js:13-181"""
    _KIND = N.RENDERER_EAM

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([
            {'name': 'extinction', 'label': 'Extinction', 'type': 'spinner', 'value': 100, 'min': 0},
            {'name': 'slices', 'label': 'Slices', 'type': 'spinner', 'value': 64, 'min': 1},
            {'name': 'random', 'label': 'Random', 'type': 'checkbox', 'value': True},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('extinction', 'slices', 'random', 'transferFunction'))   # :47-61
        self._frameNumber = 0

    def _resetFrame(self):                                         # :76-86
        N.check(N.lib().vpt_renderer_reset(self._h, None))
        self._frameNumber = 0

    def _prepare_generate(self):                                   # :99-116,120
        u = self._new_uniforms()
        u.step_size = _f32(1 / self.slices)
        u.extinction = _f32(self.extinction)
        u.offset = _f32(self.rng()) if self.random else 0.0
        self._frameNumber += 1
        self._u = u
        return u

    def _prepare_integrate(self):                                  # :135
        self._u.mix = _f32(1 / self._frameNumber)
        return self._u

    def _generateFrame(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_generate(self._h, C.byref(self._prepare_generate())))

    def _integrateFrame(self):
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._prepare_integrate())))

    def _renderFrame(self):
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def _prepare_frame_uniforms(self):
        self._prepare_generate()
        return self._prepare_integrate()

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))


class MCSRenderer(AbstractRenderer):
    """src/js/renderers/MCSRenderer.js:13-182"""
    _KIND = N.RENDERER_MCS

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([
            {'name': 'extinction', 'label': 'Extinction', 'type': 'spinner', 'value': 1, 'min': 0},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('extinction', 'transferFunction'))   # :34-47
        self._frameNumber = 1

    def _resetFrame(self):                                         # :62-72
        N.check(N.lib().vpt_renderer_reset(self._h, None))
        self._frameNumber = 1

    def _prepare_generate(self):                                   # :88-117
        u = self._new_uniforms()
        u.rand_seed = _f32(self.rng())
        u.extinction = _f32(self.extinction)
        while True:                                                # scattering direction: rejection-sampled unit vector
            x = self.rng() * 2 - 1
            y = self.rng() * 2 - 1
            z = self.rng() * 2 - 1
            length = math.sqrt(x * x + y * y + z * z)
            if not length > 1:
                break
        u.light_direction[0] = _f32(x / length)
        u.light_direction[1] = _f32(y / length)
        u.light_direction[2] = _f32(z / length)
        self._u = u
        return u

    def _prepare_integrate(self):                                  # :137-139
        self._u.mix = _f32(1 / self._frameNumber)
        self._frameNumber += 1
        return self._u

    def _generateFrame(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_generate(self._h, C.byref(self._prepare_generate())))

    def _integrateFrame(self):
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._prepare_integrate())))

    def _renderFrame(self):
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def _prepare_frame_uniforms(self):
        self._prepare_generate()
        return self._prepare_integrate()

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))


class MCMRenderer(AbstractRenderer):
    """src/js/renderers/MCMRenderer.js:13-265"""
    _KIND = N.RENDERER_MCM

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([
            {'name': 'extinction', 'label': 'Extinction', 'type': 'spinner', 'value': 1, 'min': 0},
            {'name': 'anisotropy', 'label': 'Anisotropy', 'type': 'slider', 'value': 0, 'min': -1, 'max': 1},
            {'name': 'bounces', 'label': 'Max bounces', 'type': 'spinner', 'value': 8, 'min': 0},
            {'name': 'steps', 'label': 'Steps', 'type': 'spinner', 'value': 8, 'min': 0},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('extinction', 'anisotropy', 'bounces', 'transferFunction'))   # :56-71 (not 'steps')

    def _resetFrame(self):                                         # :85-116
        u = self._new_uniforms()
        u.rand_seed = _f32(self.rng())
        u.blur = 0.0
        N.check(N.lib().vpt_renderer_reset(self._h, C.byref(u)))

    def _generateFrame(self):                                      # :118-119 (empty)
        pass

    def _prepare_integrate(self):                                  # :155-175
        u = self._new_uniforms()
        u.rand_seed = _f32(self.rng())
        u.blur = 0.0
        u.extinction = _f32(self.extinction)
        u.anisotropy = _f32(self.anisotropy)
        u.max_bounces = int(self.bounces)
        u.steps = int(self.steps)
        self._u = u
        return u

    def _integrateFrame(self):                                     # :121-185
        self._bind_volume()
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._prepare_integrate())))

    def _renderFrame(self):                                        # :187-199
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def _prepare_frame_uniforms(self):
        return self._prepare_integrate()

    def _collect_frames(self, count):
        """the same draws as `count` _prepare_integrate() calls — one rng() per frame, MCMRenderer.js:156 — without rebuilding the
        uniform block each time (the host loop of a frame sequence must stay below a 1/8 shard's 16 us of kernels)"""
        if type(self)._prepare_integrate is not MCMRenderer._prepare_integrate:
            return super()._collect_frames(count)              # a subclass changed what a frame draws
        u = self._prepare_integrate()
        seeds = [u.rand_seed] + [_f32(self.rng()) for _ in range(count - 1)]
        vars_ = np.zeros((count, 8), dtype=np.float32)
        vars_[:, 0] = seeds
        u.rand_seed = seeds[-1]                                # (the base block is the last frame's, as the generic path returns it)
        return u, vars_

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))



class ISORenderer(AbstractRenderer):
    """src/js/renderers/ISORenderer.js:13-199"""
    _KIND = N.RENDERER_ISO

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([                                  # :17-46
            {'name': 'steps', 'label': 'Steps', 'type': 'spinner', 'value': 50, 'min': 1},
            {'name': 'isovalue', 'label': 'Isovalue', 'type': 'slider', 'value': 0.5, 'min': 0, 'max': 1},
            {'name': 'light', 'label': 'Light direction', 'type': 'vector-spinner', 'value': [2, -3, -5]},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('isovalue', 'transferFunction'))      # :48-61 (steps and light do NOT reset)

    def _resetFrame(self):                                         # :75-82
        N.check(N.lib().vpt_renderer_reset(self._h, None))

    def _prepare_generate(self):                                   # :84-116
        u = self._new_uniforms()
        u.steps = int(self.steps)
        u.step_size = _f32(np.float32(1.0) / np.float32(int(self.steps)))   # the shader's 1.0 / float(uSteps), ISORenderer.glsl:64
        u.offset = _f32(self.rng())
        u.isovalue = _f32(self.isovalue)
        self._u = u
        return u

    def _prepare_render(self):                                     # :136-171
        from .scene import iso_light_direction
        light = iso_light_direction(self._camera, self._volumeTransform, self.light)
        for i in range(3):
            self._u.light_direction[i] = float(light[i])
        self._u.gradient_step = _f32(0.005)                        # :168
        return self._u

    def _generateFrame(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_generate(self._h, C.byref(self._prepare_generate())))

    def _integrateFrame(self):                                     # :118-134
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._u)))

    def _renderFrame(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render_frame(self._h, C.byref(self._prepare_render())))

    def _prepare_frame_uniforms(self):
        self._prepare_generate()
        return self._prepare_render()

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))


class DepthRenderer(AbstractRenderer):
    """src/js/renderers/DepthRenderer.js:13-191"""
    _KIND = N.RENDERER_DEPTH

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([                                  # :17-53
            {'name': 'extinction', 'label': 'Extinction', 'type': 'spinner', 'value': 100, 'min': 0},
            {'name': 'slices', 'label': 'Slices', 'type': 'spinner', 'value': 64, 'min': 1},
            {'name': 'threshold', 'label': 'Threshold', 'type': 'slider', 'value': 0.1, 'min': 0, 'max': 1},
            {'name': 'random', 'label': 'Random', 'type': 'checkbox', 'value': False},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('extinction', 'slices', 'threshold', 'random', 'transferFunction'))   # :55-70
        self._frameNumber = 0

    def _resetFrame(self):                                         # :86-95
        N.check(N.lib().vpt_renderer_reset(self._h, None))
        self._frameNumber = 0

    def _prepare_generate(self):                                   # :97-131
        u = self._new_uniforms()
        u.step_size = _f32(1 / self.slices)
        u.extinction = _f32(self.extinction)
        u.threshold = _f32(self.threshold)
        u.offset = _f32(self.rng()) if self.random else 0.0
        self._frameNumber += 1
        self._u = u
        return u

    def _prepare_integrate(self):                                  # :146
        self._u.mix = _f32(1 / self._frameNumber)
        return self._u

    def _generateFrame(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_generate(self._h, C.byref(self._prepare_generate())))

    def _integrateFrame(self):                                     # :133-149
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._prepare_integrate())))

    def _renderFrame(self):                                        # :151-163
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def _prepare_frame_uniforms(self):
        self._prepare_generate()
        return self._prepare_integrate()

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))




class LAORenderer(AbstractRenderer):
    """src/js/renderers/LAORenderer.js:13-245: emission-absorption with local ambient occlusion
    and soft shadows; each frame replaces the accumulator (LAORenderer.glsl:225-227)"""
    _KIND = N.RENDERER_LAO

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([                                  # :17-108
            {'name': 'extinction', 'label': 'Extinction', 'type': 'spinner', 'value': 100, 'min': 0},
            {'name': 'localAmbientOcclusion', 'label': 'Local Ambient Occlusion', 'type': 'checkbox', 'value': True},
            {'name': 'LAOWeight', 'label': 'LAO Weight', 'type': 'spinner', 'value': 0.69, 'min': 0, 'max': 1},
            {'name': 'numLAOSamples', 'label': '# of LAO Samples', 'type': 'spinner', 'value': 1, 'min': 1},
            {'name': 'LAOStepSize', 'label': 'LAO Stem Size', 'type': 'spinner', 'value': 0.05, 'min': 0},
            {'name': 'softShadows', 'label': 'Soft Shadows', 'type': 'checkbox', 'value': True},
            {'name': 'shadowsWeight', 'label': 'Shadows Weight', 'type': 'spinner', 'value': 0.54, 'min': 0, 'max': 1},
            {'name': 'numShadowSamples', 'label': '# of Shadow Samples', 'type': 'spinner', 'value': 10, 'min': 1},
            {'name': 'lightRadious', 'label': 'Light Radious', 'type': 'spinner', 'value': 0.19, 'min': 0},
            {'name': 'lightPosition', 'label': 'Light position', 'type': 'vector-spinner', 'value': [2, 12, 3]},
            {'name': 'lightCoeficient', 'label': 'Light Coeficient', 'type': 'spinner', 'value': 1.0, 'min': 0},
            {'name': 'slices', 'label': 'Slices', 'type': 'spinner', 'value': 64, 'min': 1},
            dict(_TF_PROPERTY),
        ])
        _install_change_handler(self, ('extinction', 'slices', 'transferFunction'))   # :110-124

    def lao_params(self):
        """the gl.uniform* block of :159-169 as struct vpt_lao_params"""
        p = N.LaoParams()
        p.local_ambient_occlusion = int(bool(self.localAmbientOcclusion))
        p.lao_weight = _f32(self.LAOWeight)
        p.num_lao_samples = int(self.numLAOSamples)
        p.lao_step_size = _f32(self.LAOStepSize)
        p.soft_shadows = int(bool(self.softShadows))
        p.shadows_weight = _f32(self.shadowsWeight)
        p.num_shadow_samples = int(self.numShadowSamples)
        p.light_radius = _f32(self.lightRadious)
        p.light_coefficient = _f32(self.lightCoeficient)
        for i in range(3):
            p.light_position[i] = _f32(self.lightPosition[i])
        return p

    def _resetFrame(self):                                         # :138-145
        N.check(N.lib().vpt_renderer_reset(self._h, None))

    def _prepare_generate(self):                                   # :147-186
        u = self._new_uniforms()
        u.step_size = _f32(1 / self.slices)
        u.extinction = _f32(self.extinction)
        u.offset = _f32(self.rng())                                # :170 draws Math.random() although the shader never reads uOffset
        N.check(N.lib().vpt_renderer_set_lao_params(self._h, C.byref(self.lao_params())))
        self._u = u
        return u

    def _generateFrame(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_generate(self._h, C.byref(self._prepare_generate())))

    def _integrateFrame(self):                                     # :188-203
        N.check(N.lib().vpt_renderer_integrate(self._h, C.byref(self._u)))

    def _renderFrame(self):                                        # :205-217
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def _prepare_frame_uniforms(self):
        return self._prepare_generate()

    def _renderFused(self):
        self._bind_volume()
        N.check(N.lib().vpt_renderer_render(self._h, C.byref(self._prepare_frame_uniforms())))


class DOSRenderer(AbstractRenderer):
    """src/js/renderers/DOSRenderer.js:14-316: directional occlusion shading — the volume is swept
    front to back in view-aligned slices, `steps` slices per render() call until the far corner is passed; every slice
    is one native pass that reads its neighbours' occlusion from the slice before"""
    _KIND = N.RENDERER_DOS

    def __init__(self, gl, volume, camera, environmentTexture, options=None):
        super().__init__(gl, volume, camera, environmentTexture, options)
        self.registerProperties([                                  # :18-63
            {'name': 'steps', 'label': 'Steps', 'type': 'spinner', 'value': 50, 'min': 1},
            {'name': 'slices', 'label': 'Slices', 'type': 'spinner', 'value': 200, 'min': 1},
            {'name': 'extinction', 'label': 'Extinction', 'type': 'spinner', 'value': 100, 'min': 0},
            {'name': 'aperture', 'label': 'Aperture', 'type': 'spinner', 'value': 30, 'min': 0, 'max': 89},
            {'name': 'samples', 'label': 'Samples', 'type': 'spinner', 'value': 8, 'min': 1, 'max': 200, 'step': 1},
            dict(_TF_PROPERTY),
        ])

        def on_samples(e):                                         # :72-74, before the reset of :76-84
            if e.detail['name'] == 'samples':
                self.generateOcclusionSamples()
        self.addEventListener('change', on_samples)
        _install_change_handler(self, ('slices', 'extinction', 'aperture', 'samples', 'transferFunction'))
        self._depth = 0
        self._minDepth = 0
        self._maxDepth = 0
        self.fused = False                                         # there is no single-launch form of this renderer
        self.generateOcclusionSamples()

    def generateOcclusionSamples(self):                            # :103-140
        from .dos_sweep import occlusion_taps
        n = int(self.samples)
        self._occlusionSamples = occlusion_taps(self.rng, n)
        N.check(N.lib().vpt_renderer_set_occlusion_samples(self._h, self._occlusionSamples.ctypes.data_as(C.c_void_p), n))

    def calculateDepth(self):                                      # :142-167
        from .dos_sweep import view_depth_range
        return view_depth_range(self._volumeTransform.globalMatrix, self._camera.transform.inverseGlobalMatrix)

    def _resetFrame(self):                                         # :169-185
        nearest, farthest = self.calculateDepth()
        self._minDepth, self._maxDepth = max(nearest, 0), farthest
        self._depth = self._minDepth
        N.check(N.lib().vpt_renderer_reset(self._h, None))

    def _prepare_slices(self):
        """the uniforms of :212-236 and, per pass of the loop :240-259, (uOcclusionScale.x, uOcclusionScale.y, uDepth)"""
        from .scene import PerspectiveCamera
        from .dos_sweep import slice_triples
        u = self._new_uniforms()
        u.extinction = _f32(self.extinction)
        sliceDistance = (self._maxDepth - self._minDepth) / self.slices
        u.step_size = _f32(sliceDistance)
        sweep = {'depth': self._depth, 'farthest': self._maxDepth}
        slices = slice_triples(sweep, int(self.steps), sliceDistance, self.aperture, self._camera.getComponent(PerspectiveCamera).projectionMatrix)
        self._depth = sweep['depth']
        self._u = u
        return u, slices

    def _integrateFrame(self):                                     # :187-262
        self._bind_volume()
        u, slices = self._prepare_slices()
        self._slices = slices
        N.check(N.lib().vpt_renderer_integrate_slices(self._h, C.byref(u), slices.ctypes.data_as(C.c_void_p), len(slices)))

    def _renderFrame(self):                                        # :264-277
        N.check(N.lib().vpt_renderer_render_frame(self._h, None))

    def play(self, count, use_graph=True, fused=False):
        raise N.VptError(N.ERR_UNSUPPORTED, 'frame sequences are not defined for the DOS renderer: drive it slice by slice')


for _cls in (MIPRenderer, EAMRenderer, MCSRenderer, MCMRenderer, ISORenderer, DepthRenderer, LAORenderer, DOSRenderer):
    _cls._BASE = _cls


def RendererFactory(which):
    """src/js/renderers/RendererFactory.js:10-23 (all eight names)."""
    classes = {'mip': MIPRenderer, 'eam': EAMRenderer, 'mcs': MCSRenderer, 'mcm': MCMRenderer,
               'iso': ISORenderer, 'depth': DepthRenderer, 'lao': LAORenderer, 'dos': DOSRenderer}
    if which not in classes:
        raise RuntimeError('No suitable class')
    return classes[which]
