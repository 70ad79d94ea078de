import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from multiviewhmr_amd import aggregation
from test_unproject_gpu import _ring_problem
gpu = torch.device('cuda:0')
for mode in ("softmax", "sum"):
    feats, proj, coords = _ring_problem(seed=57, B=4, V=4, C=4, H=400, W=400, vol=(64, 64, 32))
    f, p, c = torch.from_numpy(feats).to(gpu), torch.from_numpy(proj).to(gpu), torch.from_numpy(coords).to(gpu)
    f16 = f.half()
    o32 = aggregation.unprojection(f16, p, c, aggregation_method=mode, variant="brick", out_dtype=torch.float32)
    o16 = aggregation.unprojection(f16, p, c, aggregation_method=mode, variant="brick")
    o16b = aggregation.unprojection(f16, p, c, aggregation_method=mode, variant="brick")
    bad = (o16 != o32.half())
    print(mode, "mismatches", int(bad.sum()), "of", bad.numel(), "repeat equal", torch.equal(o16, o16b))
    idx = bad.nonzero()
    print(idx[:12].tolist())
    if len(idx):
        i = tuple(idx[0].tolist())
        print(float(o16[i]), float(o32[i]), float(o32.half()[i]))
        print("by b", bad.sum(dim=(1,2,3,4)).tolist(), "by c", bad.sum(dim=(0,2,3,4)).tolist())
        print("by x", bad.sum(dim=(0,1,3,4)).tolist()); print("by y", bad.sum(dim=(0,1,2,4)).tolist()); print("by z", bad.sum(dim=(0,1,2,3)).tolist())
