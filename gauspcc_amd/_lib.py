"""ctypes binding of libgauspcc.so (C ABI: include/gauspcc.h).

There is no CPU fallback: importing a symbol from here without the built HIP library,
or calling it without an MI355X, raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAUSPCC_LIB") or os.path.join(_HERE, "libgauspcc.so")  # env override: kernel-variant experiments (tools/)

_lib = None


class GpccError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgauspcc error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [
        ("num_points", C.c_int64),
        ("num_bytes", C.c_int64),
        ("num_levels", C.c_int32),
        ("flags", C.c_int32),
        ("level_nodes", C.c_int64 * 24),
        ("coded_nodes", C.c_int64),
        ("conv_pairs", C.c_int64),
        ("device_ms", C.c_double),
        ("ideal_bits", C.c_double),
    ]


class Profile(C.Structure):
    _fields_ = [("conv_ms", C.c_double), ("conv_launches", C.c_int64), ("conv_pair_jobs", C.c_int64),
                ("fused_ms", C.c_double), ("fused_launches", C.c_int64), ("fused_pair_jobs", C.c_int64)]


class Stage(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("bytes", C.c_double), ("brackets", C.c_int64), ("critical_ms", C.c_double)]


EXPORTS = [
    "gpcc_last_error", "gpcc_version", "gpcc_ctx_create", "gpcc_ctx_destroy", "gpcc_ctx_bytes", "gpcc_ctx_set_container_version", "gpcc_raster_order", "gpcc_voxelise",
    "gpcc_model_create", "gpcc_model_destroy", "gpcc_encode", "gpcc_decode", "gpcc_decode_to", "gpcc_encode_batch", "gpcc_decode_batch", "gpcc_sort_zyx",
    "gpcc_build_octree", "gpcc_conv3d", "gpcc_head_cdf", "gpcc_rc_encode", "gpcc_rc_decode", "gpcc_memcpy_d2d",
    "gpcc_profile_enable", "gpcc_profile_get", "gpcc_profile_stages", "gpcc_debug_trace_enable", "gpcc_debug_trace_get", "gpcc_debug_capture", "gpcc_debug_capture_get", "gpcc_debug_exclusive_scan", "gpcc_debug_launches", "gpcc_device_error_check",
    "gsac_calculate_cdf", "gsac_encode", "gsac_decode", "gsac_encode_u16", "gsac_decode_u16", "gsac_encode_const", "gsac_decode_const", "gsac_host_encode_u16", "gsac_host_decode_u16", "gsac_host_encode_f32", "gsac_host_decode_f32", "gpcc_write_files", "gpcc_read_files", "gsac_encode_gaussian", "gsac_decode_gaussian", "gsac_encode_gaussian_mixed", "gsac_decode_gaussian_mixed", "gsac_calculate_cdf_mixed", "gsac_encode_gaussian_slices", "gsac_decode_gaussian_slices", "gsac_encode_gaussian_mixed_slices", "gsac_decode_gaussian_mixed_slices", "gshac_mlp2", "gshac_mlp2_act", "gsge_forward", "gsr_visible_filter", "gsr_forward", "gsnn_generate",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C gauspcc_amd/csrc). gauspcc_amd has no CPU fallback."
        )
    # torch first: the library and torch must share ONE HIP runtime (torch ships its own libamdhip64; a process that loads
    # /opt/rocm's copy through this library before torch's ends up with two, and the second finds no device)
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    vp, i64, i32, u16 = C.c_void_p, C.c_int64, C.c_int, C.c_uint16
    L.gpcc_last_error.restype = C.c_char_p
    L.gpcc_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.gpcc_ctx_destroy.argtypes = [vp]
    L.gpcc_ctx_destroy.restype = None
    L.gpcc_ctx_bytes.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.gpcc_ctx_set_container_version.argtypes = [vp, i32]
    L.gpcc_raster_order.argtypes = [vp, vp, i32, i64, vp, vp]
    L.gpcc_voxelise.argtypes = [vp, vp, i32, i64, C.c_double, C.c_double, C.c_double, i32, vp, vp]
    L.gpcc_model_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.gpcc_model_destroy.argtypes = [vp]
    L.gpcc_model_destroy.restype = None
    L.gpcc_encode.argtypes = [vp, vp, vp, i64, i32, u16, C.POINTER(vp), C.POINTER(i64), C.POINTER(Stats), vp]
    L.gpcc_decode.argtypes = [vp, vp, vp, i64, C.POINTER(vp), C.POINTER(i64), C.POINTER(u16), C.POINTER(Stats), vp]
    L.gpcc_encode_batch.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(i64), i32, i32, C.POINTER(u16), C.POINTER(vp), C.POINTER(i64), C.POINTER(Stats), C.POINTER(i32), vp]
    L.gpcc_decode_batch.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(i64), i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(u16), C.POINTER(Stats), C.POINTER(i32), vp]
    L.gpcc_decode_to.argtypes = [vp, vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(u16), C.POINTER(Stats), vp]
    L.gpcc_sort_zyx.argtypes = [vp, vp, i64, vp, vp]
    L.gpcc_build_octree.argtypes = [vp, vp, i64, C.POINTER(i32), C.POINTER(i64), vp, vp, i64, vp]
    L.gpcc_conv3d.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, i32, vp, C.POINTER(i64), vp]
    L.gpcc_head_cdf.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.gpcc_rc_encode.argtypes = [vp, vp, i32, vp, i64, i32, C.POINTER(vp), C.POINTER(i64), vp]
    L.gpcc_rc_decode.argtypes = [vp, vp, i32, vp, i64, i64, i32, vp, vp]
    L.gpcc_memcpy_d2d.argtypes = [vp, vp, vp, i64, vp]
    L.gpcc_profile_enable.argtypes = [vp, i32]
    L.gpcc_debug_trace_enable.argtypes = [vp, i32]
    L.gpcc_debug_trace_get.argtypes = [vp, vp, vp, i32]
    L.gpcc_debug_capture.argtypes = [vp, i32]
    L.gpcc_debug_capture_get.restype = C.c_longlong
    L.gpcc_debug_capture_get.argtypes = [vp, i32, vp, C.c_longlong]
    L.gpcc_debug_exclusive_scan.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp]
    L.gpcc_debug_launches.argtypes = [i32]
    L.gpcc_device_error_check.argtypes = [vp]
    L.gpcc_debug_launches.restype = C.c_longlong
    L.gpcc_profile_get.argtypes = [vp, C.POINTER(Profile)]
    L.gpcc_profile_stages.argtypes = [vp, C.POINTER(Stage), i32, C.POINTER(i32)]
    L.gsac_calculate_cdf.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp]
    L.gsac_encode.argtypes = [vp, vp, vp, i32, i64, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), vp]
    L.gsac_decode.argtypes = [vp, vp, vp, i64, vp, i32, i64, i32, vp, vp]
    L.gsac_encode_u16.argtypes = L.gsac_encode.argtypes
    L.gsac_decode_u16.argtypes = L.gsac_decode.argtypes
    L.gsac_encode_const.argtypes = L.gsac_encode.argtypes
    L.gsac_decode_const.argtypes = L.gsac_decode.argtypes
    L.gsac_host_encode_u16.argtypes = [vp, vp, i64, i32, vp, i64, C.POINTER(i64)]
    L.gsac_host_decode_u16.argtypes = [vp, vp, i64, i64, i32, vp]
    L.gsac_host_encode_f32.argtypes = L.gsac_host_encode_u16.argtypes
    L.gsac_host_decode_f32.argtypes = L.gsac_host_decode_u16.argtypes
    L.gpcc_write_files.argtypes = [vp, vp, vp, i32, i32]
    L.gpcc_read_files.argtypes = [vp, i32, i32, C.POINTER(vp), vp]
    fp = C.POINTER(C.c_float)
    L.gsac_encode_gaussian.argtypes = [vp, vp, vp, vp, vp, i64, i32, fp, fp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), vp]
    L.gsac_decode_gaussian.argtypes = [vp, vp, vp, vp, i64, C.c_float, C.c_float, vp, i64, vp, i32, vp, vp]
    L.gsac_encode_gaussian_mixed.argtypes = [vp, vp, vp, vp, vp, i32, vp, i64, i32, fp, fp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), vp]
    L.gsac_decode_gaussian_mixed.argtypes = [vp, vp, vp, vp, i32, vp, i64, C.c_float, C.c_float, vp, i64, vp, i32, vp, vp]
    L.gsac_calculate_cdf_mixed.argtypes = [vp, vp, vp, vp, i32, vp, i64, i32, i32, vp, vp]
    L.gsac_encode_gaussian_slices.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), vp]
    L.gsac_decode_gaussian_slices.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp, i32, vp, vp]
    L.gshac_mlp2.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp]
    L.gshac_mlp2_act.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, C.c_float, vp, vp]
    L.gsac_encode_gaussian_mixed_slices.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), vp]
    L.gsac_decode_gaussian_mixed_slices.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i64, vp, i32, vp, vp]
    L.gsge_forward.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp]
    f32 = C.c_float
    L.gsr_visible_filter.argtypes = [vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, f32, f32, i32, vp, vp]
    L.gsr_forward.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, f32, f32, i32, vp, vp, C.POINTER(i64), vp]
    L.gsnn_generate.argtypes = [vp, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64), vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise GpccError(rc, lib().gpcc_last_error().decode(errors="replace"))
