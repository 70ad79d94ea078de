"""ctypes binding of include/mvhmr_unproject.h -- the only way Python reaches the kernels.

No fallback: if the shared library is missing the first call raises with the build command.
"""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libmvhmr_unproject.so")

ABI_VERSION = 4
OK, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_LAUNCH = range(5)
AGG = {"softmax": 0, "sum": 1, "mean": 2, "max": 3}
F32, F16, BF16 = 0, 1, 2
LAYOUT_BVCHW, LAYOUT_BVHWC, LAYOUT_QUAD, LAYOUT_QUAD_LOG2E = 0, 1, 2, 3
VARIANT = {"auto": 0, "gather": 1, "brick": 2}

EXPORTS = (
    "mvhmr_abi_version", "mvhmr_status_string", "mvhmr_last_error",
    "mvhmr_unproject_forward_workspace_bytes", "mvhmr_unproject_backward_workspace_bytes",
    "mvhmr_unproject_forward", "mvhmr_unproject_backward", "mvhmr_build_coord_volumes",
    "mvhmr_unproject_selected_variant", "mvhmr_preferred_layout", "mvhmr_feature_layout_bytes", "mvhmr_convert_features",
    "mvhmr_unproject_query_variant", "mvhmr_internal_lds_cache_key",
    "mvhmr_unproject_forward_cuboid", "mvhmr_unproject_backward_cuboid",
    "mvhmr_conv1x1_to_quad", "mvhmr_conv1x1_to_quad_supported", "mvhmr_conv1x1_planar", "mvhmr_conv1x1_planar_supported", "mvhmr_conv1x1_wgrad", "mvhmr_conv1x1_wgrad_supported", "mvhmr_unproject_query_variant_cuboid",
    "mvhmr_unproject_backward_supported", "mvhmr_triangulate_dlt", "mvhmr_triangulate_dlt_weighted",
    "mvhmr_unproject_forward_kernel_name",
)


class Desc(ctypes.Structure):
    """struct mvhmr_unproject_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "abi_version", "batch", "views", "channels", "feat_h", "feat_w", "vol_x", "vol_y", "vol_z",
        "method", "feat_dtype", "out_dtype", "feat_layout", "variant")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmvhmr_unproject.so is not built (%s). Build it with `python -m multiviewhmr_amd.build` "
            "(hipcc, --offload-arch=gfx950); there is no CPU fallback for the un-projection path." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32
    dp = ctypes.POINTER(Desc)
    L.mvhmr_abi_version.restype = ctypes.c_int
    L.mvhmr_status_string.restype = ctypes.c_char_p
    L.mvhmr_status_string.argtypes = [ctypes.c_int]
    L.mvhmr_last_error.restype = ctypes.c_char_p
    L.mvhmr_unproject_forward_workspace_bytes.restype = sz
    L.mvhmr_unproject_forward_workspace_bytes.argtypes = [dp]
    L.mvhmr_unproject_backward_workspace_bytes.restype = sz
    L.mvhmr_unproject_backward_workspace_bytes.argtypes = [dp]
    L.mvhmr_unproject_selected_variant.restype = ctypes.c_int
    L.mvhmr_unproject_selected_variant.argtypes = [dp]
    L.mvhmr_unproject_forward_kernel_name.restype = ctypes.c_char_p
    L.mvhmr_unproject_forward_kernel_name.argtypes = [dp]
    L.mvhmr_unproject_backward_supported.restype = ctypes.c_int
    L.mvhmr_unproject_backward_supported.argtypes = [dp]
    L.mvhmr_unproject_query_variant.restype = ctypes.c_int
    L.mvhmr_unproject_query_variant.argtypes = [dp, vp, vp, vp]
    L.mvhmr_unproject_forward.restype = ctypes.c_int
    L.mvhmr_unproject_forward.argtypes = [dp, vp, vp, vp, vp, vp, sz, vp]
    L.mvhmr_unproject_backward.restype = ctypes.c_int
    L.mvhmr_unproject_backward.argtypes = [dp, vp, vp, vp, vp, vp, vp, sz, vp]
    d3 = ctypes.POINTER(ctypes.c_double)
    L.mvhmr_unproject_forward_cuboid.restype = ctypes.c_int
    L.mvhmr_unproject_forward_cuboid.argtypes = [dp, vp, vp, vp, vp, d3, d3, vp, vp, sz, vp]
    L.mvhmr_unproject_backward_cuboid.restype = ctypes.c_int
    L.mvhmr_unproject_backward_cuboid.argtypes = [dp, vp, vp, vp, vp, vp, d3, d3, vp, vp, sz, vp]
    L.mvhmr_conv1x1_to_quad.restype = ctypes.c_int
    L.mvhmr_conv1x1_to_quad.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.mvhmr_conv1x1_to_quad_supported.restype = ctypes.c_int
    L.mvhmr_conv1x1_to_quad_supported.argtypes = [i32, i32, i32, i32]
    L.mvhmr_conv1x1_planar.restype = ctypes.c_int
    L.mvhmr_conv1x1_planar.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.mvhmr_conv1x1_planar_supported.restype = ctypes.c_int
    L.mvhmr_conv1x1_planar_supported.argtypes = [i32, i32, i32]
    L.mvhmr_conv1x1_wgrad.restype = ctypes.c_int
    L.mvhmr_conv1x1_wgrad.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.mvhmr_conv1x1_wgrad_supported.restype = ctypes.c_int
    L.mvhmr_conv1x1_wgrad_supported.argtypes = [i32, i32, i32]
    L.mvhmr_unproject_query_variant_cuboid.restype = ctypes.c_int
    L.mvhmr_unproject_query_variant_cuboid.argtypes = [dp, vp, vp, vp, d3, d3, vp]
    L.mvhmr_preferred_layout.restype = ctypes.c_int
    L.mvhmr_preferred_layout.argtypes = [dp]
    L.mvhmr_feature_layout_bytes.restype = sz
    L.mvhmr_feature_layout_bytes.argtypes = [dp, ctypes.c_int]
    L.mvhmr_convert_features.restype = ctypes.c_int
    L.mvhmr_convert_features.argtypes = [dp, vp, ctypes.c_int, vp, vp]
    L.mvhmr_internal_lds_cache_key.restype = ctypes.c_ulonglong
    L.mvhmr_internal_lds_cache_key.argtypes = [ctypes.c_int, vp]
    L.mvhmr_triangulate_dlt.restype = ctypes.c_int
    L.mvhmr_triangulate_dlt.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.mvhmr_triangulate_dlt_weighted.restype = ctypes.c_int
    L.mvhmr_triangulate_dlt_weighted.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.mvhmr_build_coord_volumes.restype = ctypes.c_int
    L.mvhmr_build_coord_volumes.argtypes = [vp, vp, vp, i32, i32, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), vp]
    if L.mvhmr_abi_version() != ABI_VERSION:
        raise RuntimeError("libmvhmr_unproject.so speaks ABI %d, this binding %d: rebuild" %
                           (L.mvhmr_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(status):
    """Map an mvhmr_status_t to the exception the reference would have raised for the same mistake."""
    if status == OK:
        return
    msg = lib().mvhmr_last_error().decode() or lib().mvhmr_status_string(status).decode()
    if status == ERR_INVALID_ARGUMENT and msg.startswith("Unknown aggregation_method"):
        raise ValueError(msg)                       # models/aggregation.py:85
    raise RuntimeError("mvhmr_unproject: %s" % msg)  # torch shape/device errors are RuntimeError too
