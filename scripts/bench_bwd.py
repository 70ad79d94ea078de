"""Times forward+backward through the autograd.Function at a given size (default: BASELINE configs[2] shape, fp32)."""
import argparse, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multiviewhmr_amd import aggregation
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--channels", type=int, default=256); ap.add_argument("--views", type=int, default=4)
ap.add_argument("--feat", type=int, default=96); ap.add_argument("--dtype", default="f32"); ap.add_argument("--iters", type=int, default=3); ap.add_argument("--variant", default="auto"); ap.add_argument("--transpose", action="store_true", help="swap the image axes (features transposed, projection rows 0 / 1 swapped): the brick windows then run along the other axis")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float32 if a.dtype == "f32" else torch.float16
f = torch.randn(a.batch, a.views, a.channels, a.feat, a.feat, device=dev).to(dt).requires_grad_(True)
P = torch.from_numpy(bench.ring_projections(a.batch, a.views, (a.feat, a.feat))).to(dev)
c = torch.from_numpy(np.ascontiguousarray(bench.cuboid_volume(1, a.grid))).to(dev).expand(a.batch, -1, -1, -1, -1).contiguous()
if a.transpose:
    f = f.detach().transpose(-1, -2).contiguous().requires_grad_(True)
    P = P[:, :, [1, 0, 2], :].contiguous()
out = aggregation.unprojection(f, P, c, variant=a.variant)
go = torch.randn_like(out)
for it in range(a.iters + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = aggregation.unprojection(f, P, c, variant=a.variant)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f.grad = None
    out.backward(go)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it: print("fwd %.2f ms  bwd %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
