"""ctypes binding of the CPU oracle (oracle/vpt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under vpt_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvpt_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, n) for n in ("vpt_oracle.c", "vpt_tonemap_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Scene(C.Structure):
    _fields_ = [
        ("volume", C.c_void_p), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
        ("filter", C.c_int32),
        ("tf_rgba", C.c_void_p), ("tf_w", C.c_int32), ("tf_h", C.c_int32),
        ("env_rgba", C.c_void_p), ("env_w", C.c_int32), ("env_h", C.c_int32),
        ("channels", C.c_int32), ("dtype", C.c_int32),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("y0", C.c_int32), ("y1", C.c_int32),
        ("mvp_inv", C.c_float * 16),
        ("seed", C.c_float), ("offset", C.c_float), ("step", C.c_float),
        ("extinction", C.c_float), ("anisotropy", C.c_float),
        ("max_bounces", C.c_uint32), ("steps", C.c_uint32),
        ("light_dir", C.c_float * 3),
        ("mix", C.c_float), ("blur", C.c_float), ("inv_res", C.c_float * 2),
        ("nthreads", C.c_int32),
        ("isovalue", C.c_float), ("gradient_step", C.c_float), ("threshold", C.c_float),
    ]


class LaoParams(C.Structure):
    """struct vpo_lao_params (oracle/vpt_oracle.c); defaults = LAORenderer.js:17-108"""
    _fields_ = [("local_ambient_occlusion", C.c_int32), ("lao_weight", C.c_float), ("num_lao_samples", C.c_int32),
                ("lao_step_size", C.c_float), ("soft_shadows", C.c_int32), ("shadows_weight", C.c_float),
                ("num_shadow_samples", C.c_int32), ("light_radius", C.c_float), ("light_coefficient", C.c_float),
                ("light_position", C.c_float * 3)]


def lao_params(**kw):
    p = LaoParams(local_ambient_occlusion=1, lao_weight=float(np.float32(0.69)), num_lao_samples=1, lao_step_size=float(np.float32(0.05)),
                  soft_shadows=1, shadows_weight=float(np.float32(0.54)), num_shadow_samples=10, light_radius=float(np.float32(0.19)),
                  light_coefficient=1.0)
    p.light_position[0], p.light_position[1], p.light_position[2] = 2.0, 12.0, 3.0
    for k, v in kw.items():
        if k == "light_position":
            for i in range(3):
                p.light_position[i] = float(np.float32(v[i]))
        elif isinstance(getattr(p, k), int):
            setattr(p, k, int(v))
        else:
            setattr(p, k, float(np.float32(v)))
    return p


class TonemapParams(C.Structure):
    """parameters of the ten tone mappers (oracle/vpt_tonemap_oracle.c); defaults = the reference's property defaults"""
    _fields_ = [("low", C.c_float), ("mid", C.c_float), ("high", C.c_float), ("saturation", C.c_float),
                ("min", C.c_float), ("max", C.c_float), ("exposure", C.c_float), ("gamma", C.c_float)]


TONEMAPPERS = ("artistic", "range", "reinhard", "reinhard2", "uncharted2", "filmic", "unreal", "aces", "lottes", "uchimura")


def tonemap_params(**kw):
    p = TonemapParams(low=0.0, mid=0.5, high=1.0, saturation=1.0, min=0.0, max=1.0, exposure=1.0, gamma=2.2)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, float(np.float32(v)))
    return p


def tonemap(kind, rgba16f, **params):
    """kind: name from TONEMAPPERS; rgba16f: [...][4] float16 -> [...][4] uint8"""
    src = np.ascontiguousarray(rgba16f, dtype=np.float16)
    out = np.empty(src.shape, dtype=np.uint8)
    p = tonemap_params(**params)
    rc = lib().vpo_tonemap(TONEMAPPERS.index(kind), C.byref(p), _ptr(src), _ptr(out), src.size // 4)
    assert rc == 0
    return out


_lib = None


def tf_rasterize(bumps, width, height, unpremultiply=True):
    """the transfer-function widget's canvas (vpo_tf_rasterize): bumps [count][8] float32 -> [height][width][4] uint8"""
    b = np.ascontiguousarray(bumps, dtype=np.float32).reshape(-1, 8)
    out = np.empty((height, width, 4), dtype=np.uint8)
    lib().vpo_tf_rasterize(_ptr(b) if len(b) else None, len(b), width, height, 1 if unpremultiply else 0, _ptr(out))
    return out


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.vpo_logf.restype = C.c_float; L.vpo_logf.argtypes = [C.c_float]
        L.vpo_sincosf.restype = None; L.vpo_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vpo_atan2f.restype = C.c_float; L.vpo_atan2f.argtypes = [C.c_float, C.c_float]
        L.vpo_asinf.restype = C.c_float; L.vpo_asinf.argtypes = [C.c_float]
        L.vpo_rcp_nr.restype = C.c_float; L.vpo_rcp_nr.argtypes = [C.c_float]
        L.vpo_rsqrt_nr.restype = C.c_float; L.vpo_rsqrt_nr.argtypes = [C.c_float]
        L.vpo_rcp_nrz.restype = C.c_float; L.vpo_rcp_nrz.argtypes = [C.c_float]
        L.vpo_sqrt_nr.restype = C.c_float; L.vpo_sqrt_nr.argtypes = [C.c_float]
        L.vpo_min.restype = C.c_float; L.vpo_min.argtypes = [C.c_float, C.c_float]
        L.vpo_max.restype = C.c_float; L.vpo_max.argtypes = [C.c_float, C.c_float]
        L.vpo_pcg.restype = C.c_uint32; L.vpo_pcg.argtypes = [C.c_uint32]
        L.vpo_hash3.restype = C.c_uint32; L.vpo_hash3.argtypes = [C.c_uint32] * 3
        L.vpo_random_uniform.restype = C.c_float; L.vpo_random_uniform.argtypes = [C.POINTER(C.c_uint32)]
        L.vpo_random_sphere.restype = None; L.vpo_random_sphere.argtypes = [C.POINTER(C.c_uint32), C.c_void_p]
        L.vpo_f32_to_f16.restype = C.c_uint16; L.vpo_f32_to_f16.argtypes = [C.c_float]
        L.vpo_srgb_to_linear.restype = C.c_float; L.vpo_srgb_to_linear.argtypes = [C.c_uint8]
        P = C.c_void_p
        SP, FP = C.POINTER(Scene), C.POINTER(Frame)
        for name in ("mip", "eam", "mcs"):
            g = getattr(L, "vpo_%s_generate" % name); g.restype = C.c_uint64; g.argtypes = [SP, FP, P]
            f = getattr(L, "vpo_%s_integrate" % name); f.restype = None; f.argtypes = [FP, P, P]
            f = getattr(L, "vpo_%s_render" % name); f.restype = None; f.argtypes = [FP, P, P]
            f = getattr(L, "vpo_%s_reset" % name); f.restype = None; f.argtypes = [FP, P]
        L.vpo_iso_generate.restype = C.c_uint64; L.vpo_iso_generate.argtypes = [SP, FP, P]
        L.vpo_iso_integrate.restype = None; L.vpo_iso_integrate.argtypes = [FP, P, P]
        L.vpo_iso_render.restype = C.c_uint64; L.vpo_iso_render.argtypes = [SP, FP, P, P]
        L.vpo_iso_reset.restype = None; L.vpo_iso_reset.argtypes = [FP, P]
        g = L.vpo_depth_generate; g.restype = C.c_uint64; g.argtypes = [SP, FP, P]
        L.vpo_depth_integrate.restype = None; L.vpo_depth_integrate.argtypes = [FP, P, P]
        L.vpo_depth_render.restype = None; L.vpo_depth_render.argtypes = [FP, P, P]
        L.vpo_depth_reset.restype = None; L.vpo_depth_reset.argtypes = [FP, P]
        L.vpo_lao_generate.restype = C.c_uint64; L.vpo_lao_generate.argtypes = [SP, FP, C.POINTER(LaoParams), P]
        L.vpo_lao_integrate.restype = None; L.vpo_lao_integrate.argtypes = [FP, P, P]
        L.vpo_lao_render.restype = None; L.vpo_lao_render.argtypes = [FP, P, P]
        L.vpo_lao_reset.restype = None; L.vpo_lao_reset.argtypes = [FP, P]
        L.vpo_dos_reset.restype = None; L.vpo_dos_reset.argtypes = [FP, P, P]
        L.vpo_dos_integrate_slice.restype = C.c_uint64
        L.vpo_dos_integrate_slice.argtypes = [SP, FP, P, P, C.c_int32, P, P, P, P]
        L.vpo_dos_render.restype = None; L.vpo_dos_render.argtypes = [FP, P, P]
        L.vpo_mcm_reset.restype = None; L.vpo_mcm_reset.argtypes = [FP, P, P, P, P]
        L.vpo_mcm_integrate.restype = C.c_uint64; L.vpo_mcm_integrate.argtypes = [SP, FP, P, P, P, P]
        L.vpo_mcm_render.restype = None; L.vpo_mcm_render.argtypes = [FP, P, P]
        L.vpo_sample_volume.restype = C.c_float; L.vpo_sample_volume.argtypes = [SP, C.c_float, C.c_float, C.c_float]
        L.vpo_sample_volume_color.restype = None; L.vpo_sample_volume_color.argtypes = [SP, C.c_float, C.c_float, C.c_float, P]
        L.vpo_sample_environment.restype = None; L.vpo_sample_environment.argtypes = [SP, C.c_float, C.c_float, C.c_float, P]
        L.vpo_unproject.restype = None; L.vpo_unproject.argtypes = [P, C.c_float, C.c_float, P, P]
        L.vpo_intersect_cube.restype = None; L.vpo_intersect_cube.argtypes = [P, P, P]
        L.vpo_max_threads.restype = C.c_int
        L.vpo_expf.restype = C.c_float; L.vpo_expf.argtypes = [C.c_float]
        L.vpo_powf.restype = C.c_float; L.vpo_powf.argtypes = [C.c_float, C.c_float]
        L.vpo_f16_to_f32.restype = C.c_float; L.vpo_f16_to_f32.argtypes = [C.c_uint16]
        L.vpo_tonemap.restype = C.c_int; L.vpo_tonemap.argtypes = [C.c_int, C.POINTER(TonemapParams), P, P, C.c_size_t]
        L.vpo_tf_rasterize.restype = C.c_int; L.vpo_tf_rasterize.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, P]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


DEFAULT_TF = np.array([[[255, 0, 0, 0], [255, 0, 0, 255]]], dtype=np.uint8)   # AbstractRenderer.js:34
DEFAULT_ENV = np.array([[[255, 255, 255, 255]]], dtype=np.uint8)              # RenderingContext.js:93


class OracleScene:
    """Holds numpy arrays alive and exposes the ctypes Scene."""

    def __init__(self, volume, filter="linear", tf=None, env=None):
        volume = np.asarray(volume)
        f32 = volume.dtype.kind == "f"                 # FLOAT / HALF_FLOAT texels: half widens to float exactly
        volume = np.ascontiguousarray(volume, dtype=np.float32 if f32 else np.uint8)
        assert volume.ndim == 3 or (volume.ndim == 4 and volume.shape[3] == 2), "volume is [z][y][x] (R8 / R32F) or [z][y][x][2] (RG8)"
        self.volume = volume
        self.tf = np.ascontiguousarray(DEFAULT_TF if tf is None else tf, dtype=np.uint8)
        self.env = np.ascontiguousarray(DEFAULT_ENV if env is None else env, dtype=np.uint8)
        assert self.tf.ndim == 3 and self.tf.shape[2] == 4
        assert self.env.ndim == 3 and self.env.shape[2] == 4
        s = Scene()
        s.volume = _ptr(self.volume)
        s.nz, s.ny, s.nx = volume.shape[:3]
        s.channels = 2 if volume.ndim == 4 else 1
        s.dtype = 1 if f32 else 0
        s.filter = 1 if filter == "linear" else 0
        s.tf_rgba = _ptr(self.tf); s.tf_h, s.tf_w = self.tf.shape[:2]
        s.env_rgba = _ptr(self.env); s.env_h, s.env_w = self.env.shape[:2]
        self.c = s


def make_frame(width, height, mvp_inv, *, seed=0.0, offset=0.0, steps=64, extinction=1.0, anisotropy=0.0,
               max_bounces=8, mcm_steps=8, light_dir=(0.0, 0.0, 1.0), mix=1.0, blur=0.0, y0=0, y1=None,
               nthreads=1, isovalue=0.5, gradient_step=0.005, threshold=0.1):
    f = Frame()
    f.width, f.height = width, height
    f.y0 = y0; f.y1 = height if y1 is None else y1
    m = np.asarray(mvp_inv, dtype=np.float32).reshape(16)
    for i in range(16):
        f.mvp_inv[i] = float(m[i])
    f.seed = float(np.float32(seed)); f.offset = float(np.float32(offset))
    f.step = float(np.float32(1.0 / steps))
    f.extinction = float(np.float32(extinction)); f.anisotropy = float(np.float32(anisotropy))
    f.max_bounces = int(max_bounces); f.steps = int(mcm_steps)
    for i in range(3):
        f.light_dir[i] = float(np.float32(light_dir[i]))
    f.mix = float(np.float32(mix)); f.blur = float(np.float32(blur))
    f.inv_res[0] = float(np.float32(1.0 / width)); f.inv_res[1] = float(np.float32(1.0 / height))
    f.nthreads = nthreads
    f.isovalue = float(np.float32(isovalue)); f.gradient_step = float(np.float32(gradient_step))
    f.threshold = float(np.float32(threshold))
    return f


class OracleRenderer:
    """Drives the oracle passes in AbstractRenderer.render() order (AbstractRenderer.js:60-76)."""

    def __init__(self, kind, scene, width, height):
        self.kind, self.scene, self.w, self.h = kind, scene, width, height
        n = width * height
        if kind == "mip":
            self.frame = np.zeros(n, np.uint8); self.acc = np.zeros(n, np.uint8)
        elif kind in ("eam", "lao"):
            self.frame = np.zeros(4 * n, np.uint8); self.acc = np.zeros(4 * n, np.uint8)
            self.lao = lao_params()
        elif kind == "mcs":
            self.frame = np.zeros(4 * n, np.float32); self.acc = np.zeros(4 * n, np.float32)
        elif kind == "iso":
            self.frame = np.zeros(4 * n, np.uint16); self.acc = np.zeros(4 * n, np.uint16)      # RGBA16F bits
        elif kind == "depth":
            self.frame = np.zeros(n, np.float32); self.acc = np.zeros(n, np.float32)            # R32F
        elif kind == "mcm":
            self.state = [np.zeros(4 * n, np.float32) for _ in range(4)]
        elif kind == "dos":                               # colour (RGBA32F) and occlusion (R32F), each double-buffered
            self.color = [np.zeros(4 * n, np.float32) for _ in range(2)]
            self.occlusion = [np.zeros(n, np.float32) for _ in range(2)]
            self.cur = 0
        else:
            raise ValueError(kind)
        self.out = np.zeros(4 * n, np.uint16)
        self.samples = 0

    def integrate_slices(self, fr, slices, samples):
        """DOSRenderer.js:240-259: one full-screen pass per row of `slices` (uOcclusionScale.xy, uDepth), ping-ponging"""
        slices = np.ascontiguousarray(slices, np.float32).reshape(-1, 3)
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1)
        for sl in slices:
            a, b = self.cur, 1 - self.cur
            self.samples += lib().vpo_dos_integrate_slice(C.byref(self.scene.c), C.byref(fr), _ptr(sl), _ptr(samples), samples.size // 2,
                                                          _ptr(self.color[a]), _ptr(self.occlusion[a]), _ptr(self.color[b]), _ptr(self.occlusion[b]))
            self.cur = b

    def reset(self, fr):
        L = lib()
        if self.kind == "dos":
            L.vpo_dos_reset(C.byref(fr), _ptr(self.color[self.cur]), _ptr(self.occlusion[self.cur]))
            return
        if self.kind == "mcm":
            L.vpo_mcm_reset(C.byref(fr), *[_ptr(s) for s in self.state])
        else:
            getattr(L, "vpo_%s_reset" % self.kind)(C.byref(fr), _ptr(self.acc))

    def generate(self, fr):
        if self.kind == "mcm":
            return 0
        if self.kind == "lao":
            n = lib().vpo_lao_generate(C.byref(self.scene.c), C.byref(fr), C.byref(self.lao), _ptr(self.frame))
        else:
            n = getattr(lib(), "vpo_%s_generate" % self.kind)(C.byref(self.scene.c), C.byref(fr), _ptr(self.frame))
        self.samples += n
        return n

    def integrate(self, fr):
        L = lib()
        if self.kind == "mcm":
            n = L.vpo_mcm_integrate(C.byref(self.scene.c), C.byref(fr), *[_ptr(s) for s in self.state])
            self.samples += n
            return n
        getattr(L, "vpo_%s_integrate" % self.kind)(C.byref(fr), _ptr(self.acc), _ptr(self.frame))
        return 0

    def render_frame(self, fr):
        L = lib()
        if self.kind == "dos":
            L.vpo_dos_render(C.byref(fr), _ptr(self.color[self.cur]), _ptr(self.out))
        elif self.kind == "mcm":
            L.vpo_mcm_render(C.byref(fr), _ptr(self.state[3]), _ptr(self.out))
        elif self.kind == "iso":                      # the ISO render pass samples the volume (gradient + material)
            self.samples += L.vpo_iso_render(C.byref(self.scene.c), C.byref(fr), _ptr(self.acc), _ptr(self.out))
        else:
            getattr(L, "vpo_%s_render" % self.kind)(C.byref(fr), _ptr(self.acc), _ptr(self.out))

    def render(self, fr):
        self.generate(fr); self.integrate(fr); self.render_frame(fr)

    def image_f16(self):
        return self.out.view(np.float16).reshape(self.h, self.w, 4)
