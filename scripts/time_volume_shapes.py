"""Forward brick kernel on volumes whose channel-plane stride is / is not a power of two (HBM channel / bank mapping of the output
stores): kernel time per Mvoxel through the C ABI (quad-planar features prepared once), batch 32, 256 ch, 4 views, 96x96 maps."""
import ctypes, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from multiviewhmr_amd import _capi
dev = torch.device("cuda:0")
L = _capi.lib(); vp = ctypes.c_void_p
B, V, C, H = 32, 4, 256, 96
feats = torch.randn(B, V, C, H, H, device=dev)
P = torch.from_numpy(bench.ring_projections(B, V, (H, H), seed=0)).to(dev)
for vol in [(64, 64, 64), (64, 56, 64), (56, 64, 64), (64, 72, 64), (72, 64, 64), (64, 64, 96), (64, 64, 32)]:
    X, Y, Z = vol
    g = np.stack(np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij"), -1).astype(np.float32)
    side = np.array([2500.0 * X / 64, 2500.0 * Y / 64, 2500.0 * Z / 64], np.float32)          # same voxel pitch as the north star
    coords = torch.from_numpy((-side / 2 + g * (side / (np.array(vol, np.float32) - 1))).astype(np.float32)).to(dev).expand(B, X, Y, Z, 3).contiguous()
    out = torch.empty(B, C, X, Y, Z, device=dev)
    d = _capi.Desc(); d.abi_version = _capi.ABI_VERSION
    d.batch, d.views, d.channels, d.feat_h, d.feat_w = B, V, C, H, H
    d.vol_x, d.vol_y, d.vol_z = X, Y, Z
    d.method, d.feat_dtype, d.out_dtype, d.feat_layout, d.variant = 0, 0, 0, _capi.LAYOUT_BVCHW, _capi.VARIANT["brick"]
    stream = vp(torch.cuda.current_stream().cuda_stream)
    conv = torch.empty(L.mvhmr_feature_layout_bytes(ctypes.byref(d), _capi.LAYOUT_QUAD), dtype=torch.uint8, device=dev)
    _capi.check(L.mvhmr_convert_features(ctypes.byref(d), vp(feats.data_ptr()), _capi.LAYOUT_QUAD, vp(conv.data_ptr()), stream))
    dk = _capi.Desc.from_buffer_copy(d); dk.feat_layout = _capi.LAYOUT_QUAD
    def run():
        _capi.check(L.mvhmr_unproject_forward(ctypes.byref(dk), vp(conv.data_ptr()), vp(P.data_ptr()), vp(coords.data_ptr()), vp(out.data_ptr()), vp(0), 0, stream))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("volume %3d x %3d x %3d  channel stride %8d B  kernel %.3f ms  %.4f ms per Mvoxel" % (X, Y, Z, X * Y * Z * 4, ms, ms / (B * X * Y * Z / 1e6)), flush=True)
    del out, coords, conv
