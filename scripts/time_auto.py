"""Times the drop-in Python path (AUTO variant, planar input: layout pass + geometry gate + kernel) fwd and bwd."""
import argparse, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multiviewhmr_amd import aggregation
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--channels", type=int, default=256); ap.add_argument("--views", type=int, default=4)
ap.add_argument("--feat", type=int, default=96); ap.add_argument("--variant", default="auto"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
f = torch.randn(a.batch, a.views, a.channels, a.feat, a.feat, device=dev).requires_grad_(True)
P = torch.from_numpy(bench.ring_projections(a.batch, a.views, (a.feat, a.feat))).to(dev)
c = torch.from_numpy(np.ascontiguousarray(bench.cuboid_volume(1, a.grid))).to(dev).expand(a.batch, -1, -1, -1, -1).contiguous()
out = aggregation.unprojection(f, P, c, variant=a.variant); go = torch.randn_like(out)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(a.iters + 1):
    ev[0].record(); out = aggregation.unprojection(f, P, c, variant=a.variant); ev[1].record()
    f.grad = None; out.backward(go); ev[2].record(); torch.cuda.synchronize()
    if it: tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print("grid %d batch %d views %d variant %s: fwd %.3f ms  bwd %.3f ms" % (a.grid, a.batch, a.views, a.variant, tf / a.iters, tb / a.iters))
