"""End-to-end VolumeGenerator.forward (the caller of the hot path) at the north-star shape with a synthetic `batch` dict:
host geometry, 1x1 conv, un-projection -- per route."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiviewhmr_amd import aggregation, multiview
B, V, C, H, S, IMG = int(os.environ.get("VOLGEN_BATCH", "32")), 4, 256, 96, 64, 384
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
packed = dict(batch)
packed["cameras_packed"] = aggregation.pack_cameras(cams, dev)
packed["keypoints_3d"] = torch.from_numpy(np.stack(batch["keypoints_3d"])).to(dev)
gen = aggregation.VolumeGenerator(volume_size=S, input_channels=C, output_channels=C, device=dev).eval()
feats = torch.randn(B, V, C, H, H, device=dev)
proj_org = torch.from_numpy(np.stack([[cams[v][b].projection for v in range(V)] for b in range(B)]).astype(np.float32)).to(dev)


def timed(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    def geometry():
        proj = torch.from_numpy(aggregation.feature_level_projections(batch["cameras"], (IMG, IMG), (H, H))).to(dev)
        return proj, gen.volume_pose(batch, proj_org, (IMG, IMG))
    def geometry_packed():
        proj = aggregation.feature_level_projections_device(packed["cameras_packed"], (IMG, IMG), (H, H))
        return proj, gen.volume_pose(packed, proj_org, (IMG, IMG))
    print("host geometry: camera objects %.3f ms | packed cameras on the device %.3f ms" % (timed(geometry), timed(geometry_packed)))
    print("1x1 conv (nn.Conv2d / MIOpen) %.2f ms" % timed(lambda: gen.process_feature(feats.view(-1, C, H, H))))
    for fused in (False, True):
        gen.fused_conv = fused
        for name, bt in (("camera objects", batch), ("packed cameras", packed)):
            print("VolumeGenerator.forward, %s, %-14s: %.2f ms" % ("fused 1x1 conv + layout" if fused else "nn.Conv2d + layout pass  ", name,
                                                                   timed(lambda: gen(feats, proj_org, bt))))
