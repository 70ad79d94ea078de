"""Per-call host overhead of the drop-in op at small shapes: wall time per unprojection() call through the C++ extension,
through ctypes, and as a HIP-graph replay of the same call (the launch floor), at the reference's shipped configuration
(cfg/defaults.py:18,23-30: 16^3 volume, 256 channels, 12 x 12 maps) and at BASELINE configs[1]."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multiviewhmr_amd import aggregation

dev = torch.device("cuda:0")
res = {}
for name, (B, S, C, HW) in {"shipped 16^3 x 256ch, 12x12 maps, batch 8": (8, 16, 256, 12), "configs[1] 32^3 x 256ch, 96x96 maps, batch 8": (8, 32, 256, 96)}.items():
    f = torch.randn(B, 4, C, HW, HW, device=dev)
    P = torch.from_numpy(bench.ring_projections(B, 4, (HW, HW))).to(dev)
    c = torch.from_numpy(np.ascontiguousarray(bench.cuboid_volume(1, S))).to(dev).expand(B, -1, -1, -1, -1).contiguous()
    row = {}
    for route, native in (("c++ extension", True), ("ctypes", False)):
        aggregation._NATIVE = native and aggregation._load_native()
        with torch.no_grad():
            for _ in range(20): aggregation.unprojection(f, P, c)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(300): aggregation.unprojection(f, P, c)
            torch.cuda.synchronize(); row[route + " us/call"] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
    aggregation._NATIVE = aggregation._load_native()
    with torch.no_grad():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): aggregation.unprojection(f, P, c)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = aggregation.unprojection(f, P, c)
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): g.replay()
        torch.cuda.synchronize(); row["HIP-graph replay us/call"] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
    res[name] = row
print(json.dumps(res, indent=1))
