"""SURVEY 8(f) row 3, training direction: VolumeGenerator -> the encoder's first stage Conv3d(256 -> 128, 3) + BatchNorm3d + ReLU
(models/regressor.py:26-32,78-80) -> backward, at batch 32 on 64^3 volumes: fp32 volume + fp32 stage against the bf16 volume
(VolumeGenerator(volume_dtype=torch.bfloat16)) + autocast stage.  One JSON line (committed as profiles/r04_consumer_stage.json)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiviewhmr_amd import aggregation, multiview
dev = torch.device("cuda:0")
B, V, C, H, S, IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4, 256, 96, 64, 384
cams = [[None] * B for _ in range(V)]
for v in range(V):
    az = 2 * np.pi * v / V + 0.3
    eye = np.array([5000 * np.cos(az), 5000 * np.sin(az), 1500.0])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    R = np.stack([right, np.cross(fwd, right), fwd])
    for b in range(B):
        cam = multiview.Camera(R, -R @ eye, [[1145.0, 0, 512], [0, 1145.0, 512], [0, 0, 1]])
        cam.update_after_crop((200, 200, 824, 824)); cam.update_after_resize((624, 624), (IMG, IMG))
        cams[v][b] = cam
batch = dict(images=np.zeros((B, V, IMG, IMG, 3), np.uint8), cameras=cams, keypoints_3d=[np.zeros((17, 3), np.float32) for _ in range(B)])
batch["cameras_packed"] = aggregation.pack_cameras(cams, dev)
proj_org = torch.from_numpy(np.stack([[cams[v][b].projection for v in range(V)] for b in range(B)]).astype(np.float32)).to(dev)
x = torch.randn(B, V, C, H, H, device=dev)
res = {"what": "VolumeGenerator (256 -> 256 conv, 64^3, 4 views, 96x96 maps) + Conv3d(256 -> 128, 3) + BatchNorm3d + ReLU, forward + backward, batch %d" % B}
for route in ("fp32", "bf16"):
    dt = torch.bfloat16 if route == "bf16" else torch.float32
    torch.manual_seed(0)
    gen = aggregation.VolumeGenerator(volume_size=S, input_channels=C, output_channels=C, device=dev, volume_dtype=dt).train()
    stage = torch.nn.Sequential(torch.nn.Conv3d(C, 128, 3, padding=1), torch.nn.BatchNorm3d(128), torch.nn.ReLU(True)).to(dev).train()
    def step(parts=None):
        xi = x.clone().requires_grad_(True)
        t = [time.perf_counter()]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=route == "bf16"):
            vol = gen(xi, proj_org, batch)
            if parts is not None: torch.cuda.synchronize(); t.append(time.perf_counter())
            y = stage(vol)
        if parts is not None: torch.cuda.synchronize(); t.append(time.perf_counter())
        y.backward(torch.ones_like(y))
        if parts is not None:
            torch.cuda.synchronize(); t.append(time.perf_counter())
            parts.append(np.diff(t))
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    parts = []
    step(parts)
    res[route] = {"ms_per_step": round(ms, 2), "aggregator_forward_ms": round(parts[0][0] * 1e3, 2), "stage_forward_ms": round(parts[0][1] * 1e3, 2),
                  "backward_ms": round(parts[0][2] * 1e3, 2), "peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    del gen, stage
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
print(json.dumps(res))
