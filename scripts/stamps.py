import ctypes, os, sys, subprocess
os.environ["MVHMR_DBG"] = "20"
sys.path.insert(0, '.')
import torch, numpy as np, bench
from multiviewhmr_amd import _capi, aggregation
L = _capi.lib()
L.mvhmr_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda:0")
B, V, C, HW, S = 32, 4, 256, 96, 64
f = torch.randn(B, V, C, HW, HW, device=dev)
P = torch.from_numpy(bench.ring_projections(B, V, (HW, HW))).to(dev)
c = torch.from_numpy(np.ascontiguousarray(bench.cuboid_volume(1, S))).to(dev).expand(B, -1, -1, -1, -1).contiguous()
out = aggregation.unprojection(f, P, c); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.mvhmr_debug_stamps(buf, 1)
out = aggregation.unprojection(f, P, c); torch.cuda.synchronize()
L.mvhmr_debug_stamps(buf, 0)
v = list(buf); n = v[7]
names = ["LDS round trip (4 reads) x4", "aggregate+store x4", "bilerp x4", "DMA issue", "vmcnt wait", "barrier", "loop total"]
for i, nm in enumerate(names):
    print("%-30s %10.0f cycles per wave per brick   %6.1f per quad   %5.1f %%" % (nm, v[i] / n, v[i] / n / 64, 100.0 * v[i] / v[6]))
print("waves", n)
