"""End-to-end VolumeGenerator.forward (the caller of the hot path: camera bookkeeping on the host, coord volumes on the
device, 1x1 conv, un-projection) at the north-star shape, with a synthetic `batch` dict.  SURVEY.md 8(d)/(f) row 1."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiviewhmr_amd import aggregation, multiview
B, V, C, H, S, IMG = 32, 4, 256, 96, 64, 384
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
cams = [[None] * B for _ in range(V)]
for v in range(V):
    az = 2 * np.pi * v / V + 0.3
    pos = np.array([5000 * np.cos(az), 5000 * np.sin(az), 1500.0])
    z = -pos / np.linalg.norm(pos); x = np.cross(z, [0, 0, 1.0]); x /= np.linalg.norm(x); y = np.cross(z, x)
    R = np.stack([x, y, z]); t = (-R @ pos).reshape(3, 1)
    K = np.array([[1145.0 * IMG / 1000, 0, IMG / 2], [0, 1145.0 * IMG / 1000, IMG / 2], [0, 0, 1.0]])
    for b in range(B):
        cams[v][b] = multiview.Camera(R, t, K)
batch = {"images": np.zeros((B, V, IMG, IMG, 3), np.uint8), "cameras": cams,
         "keypoints_3d": [rng.normal(0, 100, (17, 3)).astype(np.float32) for _ in range(B)]}
gen = aggregation.VolumeGenerator(volume_size=S, input_channels=C, output_channels=C, device=dev).eval()
feats = torch.randn(B, V, C, H, H, device=dev)
proj_org = torch.from_numpy(np.stack([[cams[v][b].projection for v in range(V)] for b in range(B)]).astype(np.float32)).to(dev)
with torch.no_grad():
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        proj = torch.from_numpy(aggregation.feature_level_projections(batch["cameras"], (IMG, IMG), (H, H))).to(dev)
        rots, centers = gen.volume_pose(batch, proj_org, (IMG, IMG))
        coords = gen.coord_volumes(rots, centers, dev)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        f2 = gen.process_feature(feats.view(-1, C, H, H)).view(B, V, C, H, H)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        vol = aggregation.unprojection(f2, proj, coords)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        vol2 = gen(feats, proj_org, batch)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        if it >= 2:
            print("geometry (host cameras + device coord volumes) %.2f ms | 1x1 conv %.2f ms | unprojection %.2f ms | forward() total %.2f ms"
                  % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
