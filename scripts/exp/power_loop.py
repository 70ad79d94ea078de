#!/usr/bin/env python3
"""A steady loop of ONE kernel family at the north-star shape for a given number of seconds (scripts/exp/power_trace.sh samples rocm-smi beside it):
power_loop.py fwd|bwd|copy <seconds>   -- fwd: mvhmr_unproject_forward on the staged copy; bwd: one backward call; copy: a 1.2-GB device copy (HBM only)"""
import sys, time, ctypes
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from multiviewhmr_amd import aggregation, _capi
what, secs = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda:0")
B, V, C, S, F = 32, 4, 256, 64, 96
f = torch.randn(B, V, C, F, F, device=dev)
P = torch.from_numpy(bench.ring_projections(B, V, (F, F))).to(dev)
c = torch.from_numpy(np.ascontiguousarray(bench.cuboid_volume(1, S))).to(dev).expand(B, -1, -1, -1, -1).contiguous()
if what == "fwd":
    L, vp = _capi.lib(), ctypes.c_void_p
    d = aggregation._make_desc(f, tuple(c.shape[1:4]), 0, torch.float32, _capi.LAYOUT_BVCHW, _capi.VARIANT["brick"])
    lay = L.mvhmr_preferred_layout(ctypes.byref(d))
    quad = torch.empty(L.mvhmr_feature_layout_bytes(ctypes.byref(d), lay) // 4, device=dev)
    st = vp(torch.cuda.current_stream(dev).cuda_stream)
    _capi.check(L.mvhmr_convert_features(ctypes.byref(d), vp(f.data_ptr()), lay, vp(quad.data_ptr()), st))
    d.feat_layout = lay
    out = torch.empty(B, C, S, S, S, device=dev)
    def step(): _capi.check(L.mvhmr_unproject_forward(ctypes.byref(d), vp(quad.data_ptr()), vp(P.data_ptr()), vp(c.data_ptr()), vp(out.data_ptr()), vp(0), 0, st))
elif what == "bwd":
    fr = f.requires_grad_(True)
    out = aggregation.unprojection(fr, P, c, variant="brick")
    go = torch.randn_like(out)
    def step():
        fr.grad = None
        torch.autograd.backward(out, go, retain_graph=True)
else:
    dst = torch.empty_like(f)
    def step(): dst.copy_(f)
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): step()
    torch.cuda.synchronize(); n += 20
print("%s: %d iterations, %.3f ms each" % (what, n, (time.perf_counter() - t0) * 1e3 / n))
