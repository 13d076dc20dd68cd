"""ctypes binding of libvpt_hip.so (C-ABI: include/vpt.h).  There is NO CPU fallback: if the HIP
library is missing or a call fails, this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VPT_HIP_LIBRARY selects another build of the same library (A/B measurements: tools/ab.sh) without touching the in-tree artefact
LIB_PATH = os.environ.get("VPT_HIP_LIBRARY") or os.path.join(_HERE, "libvpt_hip.so")

OK = 0
OPTION_MCS_PERSISTENT = 0
OPTION_MCM_PERSISTENT = 1
OPTION_FAST_MATH = 2
OPTION_BOUNDARY_ATLAS = 3
OPTION_SPLIT_STREAMS = 4
OPTION_TILE_CLASSES = 6
OPTION_VERIFY_TILE_CLASSES = 7
OPTION_BUCKET_KERNEL = 9
OPTION_COLUMN_RECORDS = 10
PLAY_EAGER, PLAY_GRAPH, PLAY_FUSED, PLAY_FRAMES = 0, 1, 2, 3
FRAME_SLOTS = 16
RENDERER_MIP, RENDERER_EAM, RENDERER_MCS, RENDERER_MCM, RENDERER_ISO, RENDERER_DEPTH, RENDERER_LAO, RENDERER_DOS = 0, 1, 2, 3, 4, 5, 6, 7
FILTER_NEAREST, FILTER_LINEAR = 0, 1
FORMAT_R8, FORMAT_RG8, FORMAT_R32F, FORMAT_RG32F = 0, 1, 2, 3
BUFFER_RENDER, BUFFER_FRAME, BUFFER_ACCUM = 0, 1, 2
BUFFER_MCM_POSITION, BUFFER_MCM_DIRECTION, BUFFER_MCM_TRANSMITTANCE, BUFFER_MCM_RADIANCE = 3, 4, 5, 6
BUFFER_DOS_OCCLUSION = 7
ERR_INVALID, ERR_HIP, ERR_NO_VOLUME, ERR_UNSUPPORTED = -1, -2, -3, -4
(PROBE_LOG, PROBE_SIN, PROBE_COS, PROBE_ASIN, PROBE_ATAN2, PROBE_PCG, PROBE_UNIFORM, PROBE_F16,
 PROBE_RCP, PROBE_RSQRT, PROBE_MIN, PROBE_MAX, PROBE_LOG_UNIFORM, PROBE_RCPZ, PROBE_SQRT, PROBE_EXP, PROBE_POW) = range(17)
TONEMAPPER_OPTION_TABLE, TONEMAPPER_TABLE_NEVER, TONEMAPPER_TABLE_ALWAYS, TONEMAPPER_TABLE_AUTO = 0, 0, 1, 2
TONEMAPPER_OPTION_FUSE = 1
(TONEMAPPER_ARTISTIC, TONEMAPPER_RANGE, TONEMAPPER_REINHARD, TONEMAPPER_REINHARD2, TONEMAPPER_UNCHARTED2, TONEMAPPER_FILMIC,
 TONEMAPPER_UNREAL, TONEMAPPER_ACES, TONEMAPPER_LOTTES, TONEMAPPER_UCHIMURA) = range(10)

# every symbol include/vpt.h declares (tests check the library exports all of them)
SYMBOLS = [
    "vpt_gather_plan", "vpt_gather_plan_recv",
    "vpt_device_count", "vpt_context_create", "vpt_context_create_on_stream", "vpt_context_destroy", "vpt_context_synchronize",
    "vpt_last_error", "vpt_version",
    "vpt_volume_create", "vpt_volume_upload_block", "vpt_volume_upload_block_device", "vpt_volume_finalize",
    "vpt_volume_set_filter", "vpt_volume_destroy", "vpt_volume_bricked_bytes", "vpt_volume_set_wide_tables",
    "vpt_renderer_create", "vpt_renderer_set_shard", "vpt_renderer_local_rows", "vpt_renderer_global_row",
    "vpt_renderer_destroy", "vpt_renderer_set_volume", "vpt_renderer_set_transfer_function",
    "vpt_renderer_set_environment", "vpt_renderer_resize",
    "vpt_renderer_reset", "vpt_renderer_generate", "vpt_renderer_integrate", "vpt_renderer_render_frame",
    "vpt_renderer_render", "vpt_renderer_play", "vpt_renderer_play_into", "vpt_renderer_play_into_display", "vpt_renderer_read", "vpt_renderer_render_buffer_device",
    "vpt_renderer_set_render_target", "vpt_renderer_join", "vpt_renderer_bucket_launches", "vpt_renderer_read_frame_slot", "vpt_renderer_frame_ring_device",
    "vpt_renderer_set_option", "vpt_renderer_set_lao_params", "vpt_renderer_set_occlusion_samples", "vpt_renderer_integrate_slices", "vpt_renderer_sample_count", "vpt_renderer_clear_sample_count",
    "vpt_renderer_set_profiling", "vpt_renderer_profile", "vpt_renderer_profile_side", "vpt_renderer_tile_classes", "vpt_classify_tiles",
    "vpt_gather_unique_id", "vpt_gather_create", "vpt_gather_destroy", "vpt_gather_render", "vpt_gather_play", "vpt_gather_set_root",
    "vpt_gather_synchronize",
    "vpt_gather_read_frame",
    "vpt_probe_math", "vpt_probe_sample", "vpt_probe_sample_boundary", "vpt_probe_stream_read", "vpt_probe_assemble_rows",
    "vpt_tonemapper_create", "vpt_tonemapper_destroy", "vpt_tonemapper_resize", "vpt_tonemapper_set_source",
    "vpt_tonemapper_set_source_image", "vpt_tonemapper_render", "vpt_tonemapper_read", "vpt_tonemapper_rows",
    "vpt_tonemapper_output_device", "vpt_tonemapper_set_option", "vpt_transfer_function_rasterize",
]


class Uniforms(C.Structure):
    """struct vpt_uniforms (include/vpt.h)"""
    _fields_ = [
        ("mvp_inverse", C.c_float * 16),
        ("rand_seed", C.c_float), ("offset", C.c_float), ("step_size", C.c_float),
        ("extinction", C.c_float), ("anisotropy", C.c_float),
        ("max_bounces", C.c_uint32), ("steps", C.c_uint32),
        ("light_direction", C.c_float * 3),
        ("mix", C.c_float), ("blur", C.c_float),
        ("isovalue", C.c_float), ("gradient_step", C.c_float), ("threshold", C.c_float), ("reserved", C.c_float),
    ]


class LaoParams(C.Structure):
    """struct vpt_lao_params (include/vpt.h)"""
    _fields_ = [("local_ambient_occlusion", C.c_int32), ("lao_weight", C.c_float), ("num_lao_samples", C.c_int32),
                ("lao_step_size", C.c_float), ("soft_shadows", C.c_int32), ("shadows_weight", C.c_float),
                ("num_shadow_samples", C.c_int32), ("light_radius", C.c_float), ("light_coefficient", C.c_float),
                ("light_position", C.c_float * 3)]


class TonemapParams(C.Structure):
    """struct vpt_tonemap_params (include/vpt.h)"""
    _fields_ = [("low", C.c_float), ("mid", C.c_float), ("high", C.c_float), ("saturation", C.c_float),
                ("min", C.c_float), ("max", C.c_float), ("exposure", C.c_float), ("gamma", C.c_float)]


class VptError(RuntimeError):
    """Non-zero return code of the C-ABI, carrying vpt_last_error() (the reference throws Error(msg))."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


_lib = None


def _share_the_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7, the same as
    /opt/rocm's).  A process that imports torch first serves this library from torch's copy — one runtime, all is well
    (bench.py, the multi-GPU hosts).  The other order leaves TWO runtimes in the process and torch then finds no GPU
    ("No HIP GPUs are available"; measured, tools/r02_exp20.sh).  So when torch is installed but not imported yet, its copy of
    the runtime is loaded first (by path, without importing torch): the order of imports stops mattering."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                     # not loadable: the library falls back to /opt/rocm's runtime


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libvpt_hip.so not found at %s — build it with `make -C vpt_amd/csrc` (or __graft_entry__.build()). "
            "vpt_amd has no CPU fallback." % LIB_PATH)
    _share_the_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    P, I, SZ = C.c_void_p, C.c_int, C.c_size_t
    PP = C.POINTER(C.c_void_p)
    UP = C.POINTER(Uniforms)
    sig = {
        "vpt_device_count": [C.POINTER(I)],
        "vpt_context_create": [I, PP], "vpt_context_create_on_stream": [I, P, PP], "vpt_context_destroy": [P], "vpt_context_synchronize": [P],
        "vpt_volume_create": [P, I, I, I, I, PP],
        "vpt_volume_upload_block": [P, I, I, I, I, I, I, P, SZ],
        "vpt_volume_upload_block_device": [P, I, I, I, I, I, I, P, SZ],
        "vpt_volume_finalize": [P], "vpt_volume_set_filter": [P, I], "vpt_volume_destroy": [P],
        "vpt_volume_bricked_bytes": [P, C.POINTER(C.c_uint64)], "vpt_volume_set_wide_tables": [P, I],
        "vpt_renderer_create": [P, I, I, I, PP],
        "vpt_renderer_set_shard": [P, I, I, I], "vpt_renderer_local_rows": [P, C.POINTER(I)],
        "vpt_renderer_global_row": [P, I, C.POINTER(I)],
        "vpt_renderer_destroy": [P], "vpt_renderer_set_volume": [P, P],
        "vpt_renderer_set_transfer_function": [P, P, I, I], "vpt_renderer_set_environment": [P, P, I, I],
        "vpt_renderer_resize": [P, I, I],
        "vpt_renderer_reset": [P, UP], "vpt_renderer_generate": [P, UP], "vpt_renderer_integrate": [P, UP],
        "vpt_renderer_render_frame": [P, UP], "vpt_renderer_render": [P, UP],
        "vpt_renderer_read": [P, I, P, SZ], "vpt_renderer_play": [P, UP, P, I, I], "vpt_renderer_play_into": [P, UP, P, I, P, SZ], "vpt_renderer_play_into_display": [P, P, UP, P, I, P, SZ],
        "vpt_gather_play": [P, UP, P, I, I], "vpt_gather_set_root": [P, I],
        "vpt_renderer_render_buffer_device": [P, PP, C.POINTER(SZ)],
        "vpt_renderer_set_render_target": [P, P, SZ], "vpt_renderer_join": [P], "vpt_renderer_bucket_launches": [P, C.POINTER(C.c_uint64)], "vpt_renderer_read_frame_slot": [P, I, P, SZ], "vpt_renderer_frame_ring_device": [P, PP, C.POINTER(SZ)],
        "vpt_renderer_set_option": [P, I, I],
        "vpt_renderer_set_lao_params": [P, C.POINTER(LaoParams)],
        "vpt_renderer_set_occlusion_samples": [P, P, I], "vpt_renderer_integrate_slices": [P, UP, P, I],
        "vpt_renderer_sample_count": [P, C.POINTER(C.c_uint64)], "vpt_renderer_clear_sample_count": [P],
        "vpt_renderer_set_profiling": [P, I],
        "vpt_renderer_tile_classes": [P, C.POINTER(I), C.POINTER(I), C.POINTER(C.c_uint64)],
        "vpt_classify_tiles": [I, I, I, I, I, P, P, SZ, C.POINTER(I), C.POINTER(I)],
        "vpt_renderer_profile": [P, C.POINTER(C.c_double), C.POINTER(C.c_uint32)],
        "vpt_renderer_profile_side": [P, C.POINTER(C.c_double), C.POINTER(C.c_uint32)],
        "vpt_probe_math": [P, I, P, P, SZ], "vpt_probe_sample": [P, P, P, SZ], "vpt_probe_sample_boundary": [P, P, P, SZ], "vpt_probe_stream_read": [P, SZ, I, P], "vpt_probe_assemble_rows": [P, P, I, I, I, I, I, P],
        "vpt_tonemapper_create": [P, I, I, I, P], "vpt_tonemapper_destroy": [P], "vpt_tonemapper_resize": [P, I, I],
        "vpt_tonemapper_set_source": [P, P], "vpt_tonemapper_set_source_image": [P, P, I, I],
        "vpt_tonemapper_render": [P, C.POINTER(TonemapParams)], "vpt_tonemapper_read": [P, P, SZ],
        "vpt_tonemapper_rows": [P, P], "vpt_tonemapper_output_device": [P, P, P], "vpt_tonemapper_set_option": [P, I, I],
        "vpt_transfer_function_rasterize": [P, P, I, I, I, I, P],
        "vpt_gather_unique_id": [P], "vpt_gather_create": [P, P, I, I, PP], "vpt_gather_destroy": [P],
        "vpt_gather_render": [P, UP], "vpt_gather_synchronize": [P], "vpt_gather_read_frame": [P, P, SZ],
    }
    for name, argtypes in sig.items():
        f = getattr(L, name)
        f.argtypes = argtypes
        f.restype = I
    L.vpt_last_error.restype = C.c_char_p; L.vpt_last_error.argtypes = []
    L.vpt_version.restype = C.c_char_p; L.vpt_version.argtypes = []
    _lib = L
    return L


def check(code):
    if code != OK:
        raise VptError(code, lib().vpt_last_error().decode("utf-8", "replace"))
