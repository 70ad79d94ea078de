"""Drop-in for the reference's models/aggregation.py on MI355X.

    unprojection(features, proj_matricies, coord_volumes, aggregation_method='softmax')   aggregation.py:20-87
    VolumeGenerator(...).forward(features, proj_matricies, batch, use_gt=True)            aggregation.py:90-195
    build_volume_generator(cfg)                                                           aggregation.py:198-208

Same names, argument meaning, return shapes/dtypes, state-dict keys and quirks (SURVEY.md section 8a,
Q1-Q6).  What differs is HOW: the b x v Python loop with ~20 ATen launches per iteration and the
(V,C,X,Y,Z) intermediates is one fused HIP kernel launch behind the C ABI of include/mvhmr_unproject.h
(forward), and one more for the gradient w.r.t. `features` (backward), exposed as torch.library custom ops
(mvhmr::unprojection / mvhmr::unprojection_cuboid + their _backward ops, with fake / meta shape functions and a
registered autograd formula) whose host side runs in the PyTorch-ROCm C++ extension csrc_ext/mvhmr_torch_ext.cpp
(or, without it, through the ctypes binding of the same C ABI).  There is no CPU / eager fallback: the call raises
if the tensors are not on a HIP device or the library is not built.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _capi, multiview, volumetric

_METHODS = ("softmax", "sum", "mean", "max")            # aggregation.py:71-85
_TYPE_MSG = "Works only with numpy arrays and PyTorch tensors."


# --------------------------------------------------------------------------------------- C-ABI plumbing
def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _workspace(nbytes, device):
    if nbytes == 0:
        return None, ctypes.c_void_p(0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)   # caching allocator: 512-B aligned, stream-ordered
    return ws, _ptr(ws)


def _is_channels_last5(features):
    """True when the (B,V,C,H,W) tensor is physically (B,V,H,W,C) -- e.g. a channels_last conv output."""
    return features.dim() == 5 and features.shape[2] > 1 and features.permute(0, 1, 3, 4, 2).is_contiguous()


def _feature_layout(features, vol, method, out_dtype, variant):
    """-> (what the library reads, its layout code, a tensor whose shape / dtype describe the features).
    A channels-last-strided tensor (physically (B,V,Hf,Wf,C)) feeds the gather kernels without a layout pass; where the library's AUTO
    choice for the same problem in planar layout is the brick kernels (3-4x the forward, 8x the backward at the north-star size) it is
    converted for them first: fp32 by the library's own channels-last -> quad-planar pass (the copy the brick kernels stage from; the
    backward then returns a planar gradient), fp16 through a planar copy."""
    if not _is_channels_last5(features):
        features = features.contiguous()
        return features, _capi.LAYOUT_BVCHW, features
    if variant == _capi.VARIANT["auto"]:
        L = _capi.lib()
        desc = _make_desc(features, vol, method, out_dtype, _capi.LAYOUT_BVCHW, variant)
        if L.mvhmr_unproject_selected_variant(ctypes.byref(desc)) == _capi.VARIANT["brick"]:
            if features.dtype != torch.float32:
                features = features.contiguous()
                return features, _capi.LAYOUT_BVCHW, features
            src = _make_desc(features, vol, method, out_dtype, _capi.LAYOUT_BVHWC, variant)
            with torch.cuda.device(features.device):
                quad = torch.empty(L.mvhmr_feature_layout_bytes(ctypes.byref(src), _capi.LAYOUT_QUAD), dtype=torch.uint8, device=features.device)
                _capi.check(L.mvhmr_convert_features(ctypes.byref(src), _ptr(features), _capi.LAYOUT_QUAD, _ptr(quad), _stream(features.device)))
            return quad, _capi.LAYOUT_QUAD, features
    return features, _capi.LAYOUT_BVHWC, features


def _dtype_code(dt):
    if dt == torch.float32:
        return _capi.F32
    if dt == torch.float16:
        return _capi.F16
    if dt == torch.bfloat16:
        return _capi.BF16                                # volume only (out_dtype with float32 features); the library checks the pairing
    raise RuntimeError("unprojection: features / volume must be float32 or float16 (volume: also bfloat16), got %s" % dt)


def _make_desc(features, coord_volumes, method, out_dtype, layout, variant):
    """coord_volumes: the (B,X,Y,Z,3) tensor, or just its (X, Y, Z)"""
    B, V, C, Hf, Wf = features.shape
    d = _capi.Desc()
    d.abi_version = _capi.ABI_VERSION
    d.batch, d.views, d.channels, d.feat_h, d.feat_w = B, V, C, Hf, Wf
    vol = tuple(coord_volumes.shape[1:4]) if torch.is_tensor(coord_volumes) else tuple(coord_volumes)
    d.vol_x, d.vol_y, d.vol_z = (int(s) for s in vol)
    d.method = method
    d.feat_dtype = _dtype_code(features.dtype)
    d.out_dtype = _dtype_code(out_dtype)
    d.feat_layout = layout
    d.variant = variant
    return d


# The op is registered with torch.library (mvhmr::unprojection / mvhmr::unprojection_backward): eager calls dispatch to the C ABI,
# FakeTensor / meta calls to the shape functions, autograd to the registered formula -- so torch.compile and AOT autograd see one
# opaque node with a known output shape and a known backward instead of a Python autograd.Function they cannot trace into.
_DTYPES = {_capi.F32: torch.float32, _capi.F16: torch.float16, _capi.BF16: torch.bfloat16}


def _load_native():
    """The PyTorch-ROCm C++ extension over the C ABI (csrc_ext/mvhmr_torch_ext.cpp, built in-tree by multiviewhmr_amd.build.build_ext):
    the per-call host work -- descriptor, output and workspace from the caching allocator, current stream, the C-ABI call -- in C++.
    Without it (not built, or MVHMR_NO_NATIVE_EXT=1) the same C-ABI calls are made through ctypes: both are the HIP path."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_ext", "mvhmr_torch_ext.so")
    if os.environ.get("MVHMR_NO_NATIVE_EXT") == "1" or not os.path.exists(path) or not os.path.exists(_capi.LIB_PATH):
        return False
    torch.ops.load_library(path)
    if torch.ops.mvhmr_native.abi_version() != _capi.ABI_VERSION:
        raise RuntimeError("mvhmr_torch_ext.so speaks ABI %d, this package %d: rebuild (python -m multiviewhmr_amd.build)"
                           % (torch.ops.mvhmr_native.abi_version(), _capi.ABI_VERSION))
    return True


_NATIVE = _load_native()


def _native_args(features, layout, like, coords, method, out_dtype, variant):
    B, V, C, Hf, Wf = like.shape
    read = features.permute(0, 1, 3, 4, 2) if layout == _capi.LAYOUT_BVHWC else features     # the contiguous view the library reads
    return (read, None, coords, B, V, C, Hf, Wf, method, _dtype_code(like.dtype), out_dtype, layout, variant)


def _op_forward(features, proj, coords, method, out_dtype, variant):
    L = _capi.lib()
    features, layout, like = _feature_layout(features, coords, method, _DTYPES[out_dtype], variant)
    if _NATIVE:
        a = _native_args(features, layout, like, coords, method, out_dtype, variant)
        return torch.ops.mvhmr_native.unprojection(a[0], proj, *a[2:])
    desc = _make_desc(like, coords, method, _DTYPES[out_dtype], layout, variant)
    B, C = like.shape[0], like.shape[2]
    with torch.cuda.device(features.device):
        out = torch.empty((B, C) + tuple(coords.shape[1:4]), dtype=_DTYPES[out_dtype], device=features.device)
        ws, wsp = _workspace(L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(desc)), features.device)
        _capi.check(L.mvhmr_unproject_forward(ctypes.byref(desc), _ptr(features), _ptr(proj), _ptr(coords), _ptr(out),
                                              wsp, 0 if ws is None else ws.numel(), _stream(features.device)))
    return out


def _op_backward(grad_out, features, proj, coords, method, out_dtype, variant):
    """gradient w.r.t. features only: proj_matricies and coord_volumes come from numpy / arange in the caller and never require grad"""
    L = _capi.lib()
    features, layout, like = _feature_layout(features, coords, method, _DTYPES[out_dtype], variant)
    grad_out = grad_out.contiguous()
    if _NATIVE:
        a = _native_args(features, layout, like, coords, method, out_dtype, variant)
        return torch.ops.mvhmr_native.unprojection_backward(grad_out, a[0], proj, *a[2:])
    desc = _make_desc(like, coords, method, _DTYPES[out_dtype], layout, variant)
    with torch.cuda.device(features.device):
        # same strides as what the library read -- except quad-planar features, whose gradient comes back planar
        grad_features = torch.empty(like.shape, dtype=like.dtype, device=like.device) if layout == _capi.LAYOUT_QUAD else torch.empty_like(features)
        ws, wsp = _workspace(L.mvhmr_unproject_backward_workspace_bytes(ctypes.byref(desc)), features.device)
        _capi.check(L.mvhmr_unproject_backward(ctypes.byref(desc), _ptr(grad_out), _ptr(features), _ptr(proj), _ptr(coords),
                                               _ptr(grad_features), wsp, 0 if ws is None else ws.numel(), _stream(features.device)))
    return grad_features


def _fake_forward(features, proj, coords, method, out_dtype, variant):
    return features.new_empty((features.shape[0], features.shape[2]) + tuple(coords.shape[1:4]), dtype=_DTYPES[out_dtype])


def _fake_backward(grad_out, features, proj, coords, method, out_dtype, variant):
    return torch.empty_like(features)


def _autograd_setup(ctx, inputs, output):
    features, proj, coords, method, out_dtype, variant = inputs
    ctx.save_for_backward(features, proj, coords)
    ctx.args = (method, out_dtype, variant)


def _autograd_backward(ctx, grad_out):
    features, proj, coords = ctx.saved_tensors
    g = torch.ops.mvhmr.unprojection_backward(grad_out, features, proj, coords, *ctx.args) if ctx.needs_input_grad[0] else None
    return g, None, None, None, None, None


def _op_defined(name):
    """True when mvhmr::<name> already exists (the module imported twice under two names, importlib.reload): define() would raise"""
    try:
        getattr(torch.ops.mvhmr, name)
        return True
    except (AttributeError, RuntimeError):
        return False


def _register_ops():
    if _op_defined("unprojection"):
        return
    sig = "(Tensor features, Tensor proj, Tensor coords, int method, int out_dtype, int variant) -> Tensor"
    torch.library.define("mvhmr::unprojection", sig)
    torch.library.define("mvhmr::unprojection_backward", "(Tensor grad_out, " + sig[1:])
    torch.library.impl("mvhmr::unprojection", "CUDA")(_op_forward)
    torch.library.impl("mvhmr::unprojection_backward", "CUDA")(_op_backward)
    torch.library.register_fake("mvhmr::unprojection")(_fake_forward)
    torch.library.register_fake("mvhmr::unprojection_backward")(_fake_backward)
    torch.library.register_autograd("mvhmr::unprojection", _autograd_backward, setup_context=_autograd_setup)


_register_ops()


def unprojection(features, proj_matricies, coord_volumes, aggregation_method='softmax', *, out_dtype=None,
                 variant='auto'):
    """Fused project -> bilinear-sample -> cross-view aggregate (reference: models/aggregation.py:20-87).

    features        (B, V, C, Hf, Wf) float32 (or float16, this package's storage mode) on a HIP device;
                    a channels-last-strided tensor (physically (B,V,Hf,Wf,C)) is consumed without a layout pass
    proj_matricies  (B, V, 3, 4)  -- feature-resolution projection matrices
    coord_volumes   (B, X, Y, Z, 3) -- voxel centres in world units
    aggregation_method  'softmax' | 'sum' | 'mean' | 'max'; anything else -> ValueError (aggregation.py:85)
    returns         a new (B, C, X, Y, Z) tensor on features.device, float32 like the reference (aggregation.py:25)
                    unless out_dtype is given (float16 features default to a float16 volume; float32 features may ask for a
                    bfloat16 volume -- what a half-precision consumer reads -- and then take a bfloat16 grad_out)

    `out_dtype` and `variant` ('auto' | 'gather' | 'brick') are keyword-only extensions.
    """
    for t in (features, proj_matricies, coord_volumes):
        if not torch.is_tensor(t):
            raise TypeError(_TYPE_MSG)                       # utils/multiview.py:110
    if aggregation_method not in _METHODS:
        raise ValueError("Unknown aggregation_method: {}".format(aggregation_method))
    if variant not in _capi.VARIANT:
        raise ValueError("Unknown kernel variant: {}".format(variant))
    if features.dim() != 5:
        raise RuntimeError("unprojection: features must be (B, V, C, Hf, Wf), got %s" % (tuple(features.shape),))
    B, V = features.shape[:2]
    if tuple(proj_matricies.shape) != (B, V, 3, 4):
        raise RuntimeError("unprojection: proj_matricies must be (%d, %d, 3, 4), got %s" % (B, V, tuple(proj_matricies.shape)))
    if coord_volumes.dim() != 5 or coord_volumes.shape[0] != B or coord_volumes.shape[4] != 3:
        raise RuntimeError("unprojection: coord_volumes must be (%d, X, Y, Z, 3), got %s" % (B, tuple(coord_volumes.shape)))
    if not features.is_cuda:
        raise RuntimeError("unprojection: features live on %s; this implementation runs only on a HIP device "
                           "(MI355X) and has no CPU path" % features.device)
    if proj_matricies.device != features.device or coord_volumes.device != features.device:
        raise RuntimeError("unprojection: expected all tensors on %s, got proj_matricies on %s and coord_volumes on %s"
                           % (features.device, proj_matricies.device, coord_volumes.device))
    if out_dtype is None:
        out_dtype = torch.float16 if features.dtype == torch.float16 else torch.float32
    if features.numel() == 0 or coord_volumes.numel() == 0:
        # empty batch / empty volume: the reference's loops do not run and its zero-initialised volume comes back
        # (aggregation.py:25-28); nothing to launch
        return torch.zeros((B, features.shape[2]) + tuple(coord_volumes.shape[1:4]), dtype=out_dtype, device=features.device)
    proj = proj_matricies.detach().to(torch.float32).contiguous()
    coords = coord_volumes.detach().to(torch.float32).contiguous()
    return torch.ops.mvhmr.unprojection(features, proj, coords, _capi.AGG[aggregation_method], _dtype_code(out_dtype), _capi.VARIANT[variant])


# The same kernels fed by the cuboid recipe instead of a coordinate tensor (mvhmr_unproject_*_cuboid), registered the same way
# (mvhmr::unprojection_cuboid / mvhmr::unprojection_cuboid_backward).
def _d3(values):
    return (ctypes.c_double * 3)(*[float(x) for x in values])


def _opc_forward(features, proj, rot, center, position, sides, vol, method, out_dtype, variant):
    L = _capi.lib()
    features, layout, like = _feature_layout(features, vol, method, _DTYPES[out_dtype], variant)
    desc = _make_desc(like, vol, method, _DTYPES[out_dtype], layout, variant)
    B, C = like.shape[0], like.shape[2]
    with torch.cuda.device(features.device):
        out = torch.empty((B, C) + tuple(vol), dtype=_DTYPES[out_dtype], device=features.device)
        ws, wsp = _workspace(L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(desc)), features.device)
        _capi.check(L.mvhmr_unproject_forward_cuboid(ctypes.byref(desc), _ptr(features), _ptr(proj), _ptr(rot), _ptr(center), _d3(position),
                                                     _d3(sides), _ptr(out), wsp, 0 if ws is None else ws.numel(), _stream(features.device)))
    return out


def _opc_backward(grad_out, features, proj, rot, center, position, sides, vol, method, out_dtype, variant):
    L = _capi.lib()
    features, layout, like = _feature_layout(features, vol, method, _DTYPES[out_dtype], variant)
    desc = _make_desc(like, vol, method, _DTYPES[out_dtype], layout, variant)
    grad_out = grad_out.contiguous()
    with torch.cuda.device(features.device):
        grad_features = torch.empty(like.shape, dtype=like.dtype, device=like.device) if layout == _capi.LAYOUT_QUAD else torch.empty_like(features)
        ws, wsp = _workspace(L.mvhmr_unproject_backward_workspace_bytes(ctypes.byref(desc)), features.device)
        _capi.check(L.mvhmr_unproject_backward_cuboid(ctypes.byref(desc), _ptr(grad_out), _ptr(features), _ptr(proj), _ptr(rot), _ptr(center),
                                                      _d3(position), _d3(sides), _ptr(grad_features), wsp, 0 if ws is None else ws.numel(),
                                                      _stream(features.device)))
    return grad_features


def _opc_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:4])
    ctx.args = tuple(inputs[4:])


def _opc_autograd(ctx, grad_out):
    features, proj, rot, center = ctx.saved_tensors
    g = torch.ops.mvhmr.unprojection_cuboid_backward(grad_out, features, proj, rot, center, *ctx.args) if ctx.needs_input_grad[0] else None
    return (g,) + (None,) * 9


def _register_cuboid_ops():
    if _op_defined("unprojection_cuboid"):
        return
    sig = "(Tensor features, Tensor proj, Tensor rot, Tensor center, float[] position, float[] sides, int[] vol, int method, int out_dtype, int variant) -> Tensor"
    torch.library.define("mvhmr::unprojection_cuboid", sig)
    torch.library.define("mvhmr::unprojection_cuboid_backward", "(Tensor grad_out, " + sig[1:])
    torch.library.impl("mvhmr::unprojection_cuboid", "CUDA")(_opc_forward)
    torch.library.impl("mvhmr::unprojection_cuboid_backward", "CUDA")(_opc_backward)
    torch.library.register_fake("mvhmr::unprojection_cuboid")(
        lambda features, proj, rot, center, position, sides, vol, method, out_dtype, variant:
        features.new_empty((features.shape[0], features.shape[2]) + tuple(vol), dtype=_DTYPES[out_dtype]))
    torch.library.register_fake("mvhmr::unprojection_cuboid_backward")(
        lambda grad_out, features, proj, rot, center, position, sides, vol, method, out_dtype, variant: torch.empty_like(features))
    torch.library.register_autograd("mvhmr::unprojection_cuboid", _opc_autograd, setup_context=_opc_setup)


_register_cuboid_ops()


def unprojection_cuboid(features, proj_matricies, rotations, centers, position, sides, volume_shape,
                        aggregation_method='softmax', *, out_dtype=None, variant='auto'):
    """`unprojection` for the volumes VolumeGenerator builds (aggregation.py:138-187), without the coordinate tensor: voxel centres
    are rot[b] @ (position + sides / (S - 1) * (i,j,k) - center[b]) + center[b], evaluated inside the kernels (bit-equal to
    mvhmr_build_coord_volumes followed by `unprojection`).

    rotations (B,3,3) and centers (B,3): float32 tensors on features.device; position, sides: 3 numbers each (cuboid corner and
    edge lengths); volume_shape: (X, Y, Z)."""
    for t in (features, proj_matricies, rotations, centers):
        if not torch.is_tensor(t):
            raise TypeError(_TYPE_MSG)
    if aggregation_method not in _METHODS:
        raise ValueError("Unknown aggregation_method: {}".format(aggregation_method))
    if variant not in _capi.VARIANT:
        raise ValueError("Unknown kernel variant: {}".format(variant))
    if features.dim() != 5:
        raise RuntimeError("unprojection: features must be (B, V, C, Hf, Wf), got %s" % (tuple(features.shape),))
    B, V = features.shape[:2]
    if tuple(proj_matricies.shape) != (B, V, 3, 4):
        raise RuntimeError("unprojection: proj_matricies must be (%d, %d, 3, 4), got %s" % (B, V, tuple(proj_matricies.shape)))
    if tuple(rotations.shape) != (B, 3, 3) or tuple(centers.shape) != (B, 3):
        raise RuntimeError("unprojection: rotations must be (%d, 3, 3) and centers (%d, 3), got %s and %s"
                           % (B, B, tuple(rotations.shape), tuple(centers.shape)))
    if not features.is_cuda:
        raise RuntimeError("unprojection: features live on %s; this implementation runs only on a HIP device "
                           "(MI355X) and has no CPU path" % features.device)
    vol = tuple(int(v) for v in volume_shape)
    if out_dtype is None:
        out_dtype = torch.float16 if features.dtype == torch.float16 else torch.float32
    if features.numel() == 0 or min(vol) == 0:
        return torch.zeros((B, features.shape[2]) + vol, dtype=out_dtype, device=features.device)
    dev = features.device
    proj = proj_matricies.detach().to(device=dev, dtype=torch.float32).contiguous()
    rot = rotations.detach().to(device=dev, dtype=torch.float32).contiguous()
    cen = centers.detach().to(device=dev, dtype=torch.float32).contiguous()
    return torch.ops.mvhmr.unprojection_cuboid(features, proj, rot, cen, [float(x) for x in position], [float(x) for x in sides], list(vol),
                                               _capi.AGG[aggregation_method], _dtype_code(out_dtype), _capi.VARIANT[variant])


# --------------------------------------------------------------------------------------- caller side
def feature_level_projections(cameras, images_shape, features_shape):
    """(B, V, 3, 4) float32 numpy: projection matrices at feature-map resolution.

    Reference: deep-copies every Camera, calls update_after_resize(images_shape, features_shape) and reads
    .projection (aggregation.py:127-133, utils/multiview.py:33-52) -- including quirk Q3 (features_shape
    = (Hf, Wf) is unpacked as (new_width, new_height)).  Here the same float64 arithmetic runs on copies
    of K only; `cameras` is list[V] of list[B] and is not modified.
    """
    height, width = images_shape
    new_width, new_height = features_shape
    sx, sy = new_width / width, new_height / height
    V, B = len(cameras), len(cameras[0])
    # one pass over the Python objects, then batched float64 math (same operations, same order as Camera does them)
    K = np.array([[cameras[v][b].K for v in range(V)] for b in range(B)], dtype=np.float64)          # (B,V,3,3), a copy
    Rt = np.array([[np.hstack([cameras[v][b].R, cameras[v][b].t]) for v in range(V)] for b in range(B)], dtype=np.float64)
    K[..., 0, 0] = K[..., 0, 0] * sx
    K[..., 1, 1] = K[..., 1, 1] * sy
    K[..., 0, 2] = K[..., 0, 2] * sx
    K[..., 1, 2] = K[..., 1, 2] * sy
    # K.dot([R|t]) per camera: spelled as the same k-ordered sum of products numpy's 3x3 @ 3x4 dot performs
    P = K[..., :, 0:1] * Rt[..., 0:1, :]
    P = P + K[..., :, 1:2] * Rt[..., 1:2, :]
    P = P + K[..., :, 2:3] * Rt[..., 2:3, :]
    return P.astype(np.float32)


class _FusedAggregate(torch.autograd.Function):
    """process_feature (1x1 conv) + un-projection with the conv output living only in the quad-planar layout the brick forward
    stages (mvhmr_conv1x1_to_quad + mvhmr_unproject_forward_cuboid on MVHMR_LAYOUT_QUAD): the planar (B,V,C,Hf,Wf) conv output
    and the layout pass over it are never written.  The un-projection runs with MVHMR_VARIANT_AUTO: the
    geometry gate decides brick / gather on the device for THIS call's cameras and pose (the gather side converts the copy to
    channels-last first), forward and backward each for their own bricks -- no cached decision, no host synchronisation.
    Backward: the un-projection backward gives the gradient w.r.t. the conv output in the planar layout; weight / bias / input
    gradients are three GEMMs on it."""

    @staticmethod
    def forward(ctx, x, weight, bias, proj, rot, center, position, sides, vol, method, out_dtype=torch.float32):
        L = _capi.lib()
        B, V, Cin, Hf, Wf = x.shape
        Cout = weight.shape[0]
        dev = x.device
        x = x.contiguous()
        w2 = weight.reshape(Cout, Cin).contiguous()
        pos = (ctypes.c_double * 3)(*[float(v) for v in position])
        sid = (ctypes.c_double * 3)(*[float(v) for v in sides])
        with torch.cuda.device(dev):
            quad = torch.empty(B * V * Cout * Hf * Wf, dtype=torch.float32, device=dev)
            _capi.check(L.mvhmr_conv1x1_to_quad(_ptr(x), _ptr(w2), _ptr(bias) if bias is not None else ctypes.c_void_p(0), _ptr(quad),
                                                B * V, Cin, Cout, Hf, Wf, _stream(dev)))
            desc = _make_desc(torch.empty((B, V, Cout, Hf, Wf), dtype=torch.float32, device="meta"), vol, method, out_dtype, _capi.LAYOUT_QUAD, _capi.VARIANT["auto"])
            out = torch.empty((B, Cout) + tuple(vol), dtype=out_dtype, device=dev)
            ws, wsp = _workspace(L.mvhmr_unproject_forward_workspace_bytes(ctypes.byref(desc)), dev)
            _capi.check(L.mvhmr_unproject_forward_cuboid(ctypes.byref(desc), _ptr(quad), _ptr(proj), _ptr(rot), _ptr(center), pos, sid,
                                                         _ptr(out), wsp, 0 if ws is None else ws.numel(), _stream(dev)))
        ctx.save_for_backward(x, w2, quad, proj, rot, center)
        ctx.desc, ctx.pos, ctx.sid, ctx.has_bias, ctx.wshape = desc, pos, sid, bias is not None, tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, w2, quad, proj, rot, center = ctx.saved_tensors
        L = _capi.lib()
        desc = ctx.desc
        B, V, Cin, Hf, Wf = x.shape
        Cout = w2.shape[0]
        dev = x.device
        grad_out = grad_out.contiguous()
        with torch.cuda.device(dev):
            gy = torch.empty((B * V, Cout, Hf * Wf), dtype=torch.float32, device=dev)           # gradient w.r.t. the conv output, planar
            ws, wsp = _workspace(L.mvhmr_unproject_backward_workspace_bytes(ctypes.byref(desc)), dev)
            _capi.check(L.mvhmr_unproject_backward_cuboid(ctypes.byref(desc), _ptr(grad_out), _ptr(quad), _ptr(proj), _ptr(rot), _ptr(center),
                                                          ctx.pos, ctx.sid, _ptr(gy), wsp, 0 if ws is None else ws.numel(), _stream(dev)))
        xf = x.view(B * V, Cin, Hf * Wf)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if L.mvhmr_conv1x1_planar_supported(Cout, Cin, Hf * Wf):                           # (Cin, Cout) @ (BV, Cout, HW) on the MFMA GEMM
                gx = torch.empty((B, V, Cin, Hf, Wf), dtype=torch.float32, device=dev)
                wt = w2.t().contiguous()
                with torch.cuda.device(dev):
                    _capi.check(L.mvhmr_conv1x1_planar(_ptr(gy), _ptr(wt), ctypes.c_void_p(0), _ptr(gx), B * V, Cout, Cin, Hf * Wf, _stream(dev)))
            else:
                gx = torch.matmul(w2.t(), gy).view(B, V, Cin, Hf, Wf)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and L.mvhmr_conv1x1_wgrad_supported(Cin, Cout, Hf * Wf):
            gw = torch.zeros(ctx.wshape, dtype=torch.float32, device=dev)                      # split-K GEMM adds into it
            gb = torch.zeros(Cout, dtype=torch.float32, device=dev) if want_b else None
            with torch.cuda.device(dev):
                _capi.check(L.mvhmr_conv1x1_wgrad(_ptr(gy), _ptr(x), _ptr(gw), _ptr(gb) if want_b else ctypes.c_void_p(0), B * V, Cin, Cout,
                                                  Hf * Wf, _stream(dev)))
        else:
            if ctx.needs_input_grad[1]:
                gw = torch.einsum("nop,nip->oi", gy, xf).view(ctx.wshape)
            if want_b:
                gb = gy.sum(dim=(0, 2))
        return gx, gw, gb, None, None, None, None, None, None, None, None


def pack_cameras(cameras, device):
    """One pass over batch['cameras'] (list[V] of list[B] of Camera) -> float64 device tensors K (B,V,3,3) and Rt (B,V,3,4).
    A data loader that hands `batch['cameras_packed'] = pack_cameras(...)` (or builds the two tensors itself) lets
    VolumeGenerator.forward derive the feature-level projection matrices on the device: no Python loop over B x V cameras and no
    host-to-device copy per call."""
    V, B = len(cameras), len(cameras[0])
    K = np.array([[cameras[v][b].K for v in range(V)] for b in range(B)], dtype=np.float64)
    Rt = np.array([[np.hstack([cameras[v][b].R, cameras[v][b].t]) for v in range(V)] for b in range(B)], dtype=np.float64)
    return {"K": torch.from_numpy(K).to(device), "Rt": torch.from_numpy(Rt).to(device)}


def feature_level_projections_device(packed, images_shape, features_shape):
    """feature_level_projections on the device from packed cameras: the same float64 operations in the same order
    (update_after_resize incl. quirk Q3, then K @ [R|t] as a k-ordered sum of products), cast to float32 at the end."""
    height, width = images_shape
    new_width, new_height = features_shape
    sx, sy = new_width / width, new_height / height
    K = packed["K"].clone()
    Rt = packed["Rt"]
    K[..., 0, 0] = K[..., 0, 0] * sx
    K[..., 1, 1] = K[..., 1, 1] * sy
    K[..., 0, 2] = K[..., 0, 2] * sx
    K[..., 1, 2] = K[..., 1, 2] * sy
    P = K[..., :, 0:1] * Rt[..., 0:1, :]
    P = P + K[..., :, 1:2] * Rt[..., 1:2, :]
    P = P + K[..., :, 2:3] * Rt[..., 2:3, :]
    return P.to(torch.float32)


class VolumeGenerator(nn.Module):
    """1x1 conv on the per-view feature maps + un-projection into a voxel volume (aggregation.py:90-195)."""

    def __init__(self, volume_size=64, input_channels=256, output_channels=32, cuboid_side=2500.0,
                 aggregation_method='softmax', use_triangulation=False, kind='mpii', device='cuda',
                 dataset='human36m', volume_dtype=None, **kwargs):
        # **kwargs swallows unknown keywords exactly like the reference (quirk Q5: build_volume_generator
        # passes volume_aggregation_method=, so aggregation_method keeps its default)
        super().__init__()
        self.volume_size = volume_size
        self.cuboid_side = cuboid_side
        self.aggregation_method = aggregation_method
        self.process_feature = nn.Sequential(nn.Conv2d(input_channels, output_channels, 1))   # keys process_feature.0.*
        self.use_triangulation = use_triangulation
        self.kind = kind
        self.dataset = dataset
        self.volume_dtype = volume_dtype  # extension: None = float32 like the reference; torch.bfloat16 / float16 for a half-precision consumer
        self.fused_conv = True            # 1x1 conv + layout pass as one MFMA GEMM wherever its shapes allow (see _fused_path_applies)
        self.to(device)

    # -- geometry the reference builds inside forward(); split out so it can be checked without a GPU
    def cuboid(self):
        sides = np.array([self.cuboid_side, self.cuboid_side, self.cuboid_side])
        position = np.array([0, 0, 0]) - sides / 2          # quirk Q4: centred on the world origin (:140-144)
        return volumetric.Cuboid3D(position, sides)

    def rotation_axis(self):
        if self.kind == "coco":
            return [0, 1, 0]
        if self.kind == "mpii":
            return [0, 0, 1]
        raise ValueError("Unknown kind: {}".format(self.kind))  # the reference fails with UnboundLocalError here

    def volume_pose(self, batch, proj_matricies_org, images_shape):
        """Per-sample rotation (B,3,3) and pivot (B,3), float32 numpy/tensor (aggregation.py:163-181).

        Training draws theta ~ U(0, 2 pi) from the GLOBAL numpy stream, one draw per sample in order
        (quirk Q6); eval uses theta = 0.  The pivot is keypoints_3d[b][6, :3], or the DLT-triangulated
        image centre when use_triangulation is set."""
        batch_size = proj_matricies_org.shape[0]
        axis = self.rotation_axis()
        if self.training:
            # one draw per sample, in order, from the GLOBAL numpy stream (Q6): a sized draw consumes the same stream
            thetas = np.random.uniform(0.0, 2 * np.pi, size=batch_size)
            rots = volumetric.get_rotation_matrices(axis, thetas).astype(np.float32)
        else:
            # theta = 0 for every sample: the same matrices call after call -- kept on the device, so that an eval forward with packed
            # cameras and tensor keypoints copies nothing from the host (and can be captured into a HIP graph: scripts/graph_volgen.py)
            key = (batch_size, proj_matricies_org.device)
            cache = self.__dict__.setdefault("_eval_rots", {})
            if key not in cache:
                r0 = np.broadcast_to(volumetric.get_rotation_matrix(axis, 0.0).astype(np.float32), (batch_size, 3, 3))
                cache[key] = torch.from_numpy(np.array(r0, dtype=np.float32)).to(proj_matricies_org.device)
            rots = None
        if self.use_triangulation:
            # one batched DLT on the device, no per-sample .cpu(); stays a device tensor
            n_views = proj_matricies_org.shape[1]
            images_center = (torch.tensor(images_shape, dtype=torch.float32) / 2).expand(n_views, 2)
            centers = multiview.triangulate_points_from_multiple_views_linear_batch(proj_matricies_org.detach(), images_center)
        else:
            kp = batch['keypoints_3d']
            if torch.is_tensor(kp):                                          # already a (B, 17, 3|4) tensor (any device)
                centers = kp[:, 6, :3].to(torch.float32)
            else:
                centers = torch.from_numpy(np.stack([np.asarray(kp[b][6, :3], dtype=np.float32) for b in range(batch_size)]))
        if rots is None:
            return cache[key], centers
        return torch.from_numpy(np.ascontiguousarray(rots)), centers

    def coord_volumes(self, rots, centers, device):
        """(B,S,S,S,3) float32 on `device`: rot @ (grid - center) + center, built by one kernel
        (mvhmr_build_coord_volumes) instead of B x (meshgrid + 3 strided writes + mm) (aggregation.py:138-187)."""
        L = _capi.lib()
        B, S = rots.shape[0], self.volume_size
        cub = self.cuboid()
        rots = rots.to(device=device, dtype=torch.float32).contiguous()
        centers = centers.to(device=device, dtype=torch.float32).contiguous()
        with torch.cuda.device(device):
            coords = torch.empty(B, S, S, S, 3, dtype=torch.float32, device=device)
            pos = (ctypes.c_double * 3)(*[float(x) for x in cub.position])
            sides = (ctypes.c_double * 3)(*[float(x) for x in cub.sides])
            _capi.check(L.mvhmr_build_coord_volumes(_ptr(coords), _ptr(rots), _ptr(centers), B, S, pos, sides, _stream(device)))
        return coords

    def forward(self, features, proj_matricies, batch, use_gt=True):
        features_shape = tuple(features.shape[-2:])
        images_shape = tuple(batch['images'].shape[2:-1])
        batch_size, n_views = batch['images'].shape[:2]
        device = features.device

        proj_org = proj_matricies                                           # only read (reference clones, :124)
        if 'cameras_packed' in batch:                                       # device tensors: no camera loop, no H2D copy
            proj = feature_level_projections_device(batch['cameras_packed'], images_shape, features_shape)
        else:
            proj = torch.from_numpy(feature_level_projections(batch['cameras'], images_shape, features_shape))
        # the kernels take raw device pointers: whatever device the packed cameras live on (a loader may pack them on the host)
        proj = proj.to(device=device, dtype=torch.float32).contiguous()
        rots, centers = self.volume_pose(batch, proj_org, images_shape)
        cub = self.cuboid()
        S = self.volume_size
        rots = rots.to(device=device, dtype=torch.float32).contiguous()
        centers = centers.to(device=device, dtype=torch.float32).contiguous()

        if self._fused_path_applies(features, S):
            # 1x1 conv and layout pass in one MFMA GEMM, its output only ever exists in the layout the brick forward stages
            conv = self.process_feature[0]
            return _FusedAggregate.apply(features, conv.weight, conv.bias, proj, rots, centers, tuple(cub.position), tuple(cub.sides),
                                         (S, S, S), _capi.AGG[self.aggregation_method], self.volume_dtype or torch.float32)

        features = features.view(-1, *features.shape[2:])
        features = self.process_feature(features)
        features = features.view(batch_size, n_views, *features.shape[1:])
        # the coordinate volumes (aggregation.py:138-187) are never materialised: the kernels evaluate the cuboid recipe per voxel
        return unprojection_cuboid(features, proj, rots, centers, cub.position, cub.sides, (S, S, S),
                                   aggregation_method=self.aggregation_method, out_dtype=self.volume_dtype)

    def _fused_path_applies(self, features, S):
        """The fused conv writes the quad-planar layout, which the un-projection consumes for every geometry (brick kernels as it
        is, gather kernels through one conversion; the device-side gate picks per call).  So only shapes, dtypes and devices
        decide here -- checked on every call, nothing cached: fp32 everywhere, conv parameters on the features' device, a shape the
        fused GEMM takes and one whose quad-planar copy the un-projection (forward and backward) accepts."""
        if not self.fused_conv or not features.is_cuda or features.dtype != torch.float32 or self.aggregation_method not in _METHODS:
            return False
        if self.volume_dtype not in (None, torch.float32, torch.bfloat16):
            return False                                                          # fp32 conv output with an fp16 volume: not a storage mode of the library
        conv = self.process_feature[0]
        params = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
        if any(t.dtype != torch.float32 or t.device != features.device for t in params):
            return False
        B, V, Cin, Hf, Wf = features.shape
        Cout = conv.out_channels
        L = _capi.lib()
        if not L.mvhmr_conv1x1_to_quad_supported(Cin, Cout, Hf, Wf):
            return False
        meta = torch.empty((B, V, Cout, Hf, Wf), dtype=torch.float32, device="meta")
        desc = _make_desc(meta, (S, S, S), _capi.AGG[self.aggregation_method], self.volume_dtype or torch.float32, _capi.LAYOUT_QUAD, _capi.VARIANT["auto"])
        return L.mvhmr_unproject_selected_variant(ctypes.byref(desc)) > 0 and L.mvhmr_unproject_backward_supported(ctypes.byref(desc)) == 1


def build_volume_generator(cfg):
    """cfg is the reference's yacs tree (cfg/defaults.py:18-30,89-90); same wiring as aggregation.py:198-208."""
    input_channels = cfg.MODEL.BACKBONE.DECONV_FILTERS[-1] if cfg.MODEL.BACKBONE.DECONV_LAYERS != 0 else 2048
    return VolumeGenerator(volume_size=cfg.MODEL.AGGREGATION.VOLUME_SIZE,
                           input_channels=input_channels,
                           output_channels=cfg.MODEL.AGGREGATION.OUTPUT_CHANNELS,
                           cuboid_side=cfg.MODEL.AGGREGATION.CUBOID_SIDE,
                           use_triangulation=cfg.MODEL.AGGREGATION.USE_TRIANGULATION,
                           kind=cfg.DATASET.KIND,
                           dataset=cfg.DATASET.TYPE,
                           volume_aggregation_method=cfg.MODEL.AGGREGATION.METHOD)
