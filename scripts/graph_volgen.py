"""VolumeGenerator.forward (eval, packed cameras: all geometry on the device) captured into a HIP graph: replay against the eager call,
and the latency of both at batch 1 (the serving case).  usage (GPU box): python scripts/graph_volgen.py [batch]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiviewhmr_amd import aggregation, multiview
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
V, C, H, S, IMG = 4, 256, 96, 64, 384
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
         "keypoints_3d": torch.from_numpy(np.stack([rng.normal(0, 100, (17, 3)).astype(np.float32) for _ in range(B)])).to(dev)}
batch["cameras_packed"] = aggregation.pack_cameras(cams, dev)
gen = aggregation.VolumeGenerator(volume_size=S, input_channels=C, output_channels=C, device=dev).eval()
feats = torch.randn(B, V, C, H, H, device=dev)
proj_org = torch.from_numpy(np.stack([[cams[v][b].projection for v in range(V)] for b in range(B)]).astype(np.float32)).to(dev)
with torch.no_grad():
    eager = gen(feats, proj_org, batch).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): gen(feats, proj_org, batch)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = gen(feats, proj_org, batch)
    g.replay(); torch.cuda.synchronize()
    print("graph replay == eager:", bool(torch.equal(out, eager)), "max diff", float((out - eager).abs().max()))
    def timed(fn, n=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print("batch %d: eager %.3f ms | graph replay %.3f ms" % (B, timed(lambda: gen(feats, proj_org, batch)), timed(g.replay)))
