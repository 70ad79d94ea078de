#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Only ever run in the build container (the reference lives at /root/reference and
never travels).  The script imports the reference's hot path unmodified
(`models.aggregation`, `utils.multiview`, `utils.volumetric`); the only accommodation
is an empty `cv2` module object, because `utils/volumetric.py:2` imports cv2 for its
drawing helpers, which nothing on this path calls.

What is written (all float32 unless noted; every file holds inputs AND expected outputs):

  unproj_<case>.npz   features, proj, coords -> out_<mode> for the four aggregation modes,
                      grad_out (fixed, seeded) -> gfeat_<mode>  (autograd through the reference)
  volgen_<case>.npz   synthetic `batch` (camera K/R/t, keypoints) + conv weights ->
                      the proj matrices / coord volumes / post-conv features the reference's
                      VolumeGenerator.forward hands to unprojection, and its returned volume
  geometry.npz        Camera bookkeeping, homogeneous helpers, rotation matrix, DLT triangulation

Usage:  python tests/golden/make_golden.py        (needs /root/reference; CPU only)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("MVHMR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch

if "cv2" not in sys.modules:
    sys.modules["cv2"] = types.ModuleType("cv2")  # drawing-only dependency, never called here
sys.path.insert(0, REF)
from models import aggregation as ref_agg  # noqa: E402
from utils import multiview as ref_mv  # noqa: E402
from utils import volumetric as ref_vol  # noqa: E402

MODES = ("softmax", "sum", "mean", "max")
torch.set_num_threads(4)


# --------------------------------------------------------------------------- synthetic geometry
def ring_cameras(n_views, radius, height, focal, sensor, rng, jitter=0.0):
    """n_views pin-hole cameras on a ring, z-up world, looking at the origin (mm units)."""
    cams = []
    for k in range(n_views):
        az = 2.0 * np.pi * k / n_views + 0.3
        r = radius * (1.0 + jitter * rng.uniform(-1, 1))
        eye = np.array([r * np.cos(az), r * np.sin(az), height * (1.0 + jitter * rng.uniform(-1, 1))])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=0)
        t = -R @ eye
        K = np.array([[focal, 0.0, sensor / 2.0], [0.0, focal, sensor / 2.0], [0.0, 0.0, 1.0]])
        cams.append((R, t, K))
    return cams


def feature_level_projections(cams, bbox, image_hw, feat_hw):
    """reference bookkeeping: crop -> resize to image_hw -> resize to feat_hw (Q3 included)."""
    out = []
    for R, t, K in cams:
        cam = ref_mv.Camera(R, t, K)
        cam.update_after_crop(bbox)
        side_h, side_w = bbox[3] - bbox[1], bbox[2] - bbox[0]
        cam.update_after_resize((side_h, side_w), (image_hw[1], image_hw[0]))
        cam.update_after_resize(image_hw, feat_hw)
        out.append(cam.projection)
    return np.stack(out).astype(np.float32)


def cuboid_coords(shape, side, center=(0.0, 0.0, 0.0), theta=0.0):
    X, Y, Z = shape
    gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    g = np.stack([gx, gy, gz], -1).astype(np.float64)
    pos = -side / 2.0
    steps = np.array([side / max(X - 1, 1), side / max(Y - 1, 1), side / max(Z - 1, 1)])
    c = pos + g * steps
    rot = ref_vol.get_rotation_matrix([0, 0, 1], theta)
    ctr = np.asarray(center, dtype=np.float64)
    c = (c - ctr) @ rot.T + ctr
    return c.astype(np.float32)


# --------------------------------------------------------------------------- unprojection cases
ONLY = set(a for a in os.environ.get("MVHMR_GOLDEN_ONLY", "").split(",") if a)   # regenerate only these unprojection cases


def run_unprojection_case(name, features, proj, coords, seed, modes=MODES, grad_modes=MODES):
    if ONLY and name not in ONLY:
        return                      # (the shared rng has been advanced by the caller either way: cases stay reproducible)
    feats = torch.from_numpy(features)
    P = torch.from_numpy(proj)
    Cv = torch.from_numpy(coords)
    g = torch.Generator().manual_seed(seed + 1000)
    B, V, C = features.shape[:3]
    grad_out = torch.randn(B, C, *coords.shape[1:4], generator=g)
    rec = dict(features=features, proj=proj, coords=coords, grad_out=grad_out.numpy())
    for mode in modes:
        f = feats.clone().requires_grad_(mode in grad_modes)
        out = ref_agg.unprojection(f, P, Cv, aggregation_method=mode)
        rec["out_" + mode] = out.detach().numpy()
        if mode in grad_modes:
            (out * grad_out).sum().backward()
            rec["gfeat_" + mode] = f.grad.numpy()
    path = os.path.join(HERE, "unproj_%s.npz" % name)
    np.savez_compressed(path, **rec)
    print("wrote", path, {k: v.shape for k, v in rec.items() if k.startswith("out_s")})


def unprojection_cases():
    rng = np.random.default_rng(7)

    def feats(B, V, C, H, W, seed, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(B, V, C, H, W, generator=g) * scale).numpy()

    # 1. tiny, well-behaved: every voxel in front of every camera, most inside the frame
    cams = ring_cameras(2, 5000.0, 1500.0, 1145.0, 1000.0, rng)
    P = feature_level_projections(cams, (150, 150, 850, 850), (64, 64), (12, 12))
    proj = np.stack([P, P[::-1]])  # B = 2
    coords = np.stack([cuboid_coords((5, 6, 7), 2500.0), cuboid_coords((5, 6, 7), 2500.0, theta=0.7)])
    run_unprojection_case("tiny_b2v2c4", feats(2, 2, 4, 12, 12, 1), proj, coords, 1)

    # 2. non-square feature map (exposes Q1), 3 views
    cams = ring_cameras(3, 4800.0, 1400.0, 1145.0, 1000.0, rng, jitter=0.05)
    P = feature_level_projections(cams, (100, 180, 900, 820), (96, 120), (20, 28))
    coords = cuboid_coords((6, 5, 8), 2400.0, center=(30.0, -50.0, 80.0), theta=1.1)[None]
    run_unprojection_case("nonsquare_v3c5", feats(1, 3, 5, 20, 28, 2), P[None], coords, 2)

    # 3. adversarial: camera INSIDE the cuboid (voxels behind it), frustum misses part of the grid,
    #    and one voxel placed exactly on a camera's principal plane (z == 0 -> guard path)
    cams = ring_cameras(4, 900.0, 200.0, 700.0, 1000.0, rng)
    P = feature_level_projections(cams, (200, 200, 800, 800), (64, 64), (16, 16))
    coords = cuboid_coords((7, 7, 6), 2500.0)
    R0, t0, _ = cams[0]
    eye0 = -R0.T @ t0
    coords[0, 0, 0] = eye0.astype(np.float32)          # projects to (0,0,0) up to rounding
    proj = P[None].copy()
    coords = coords[None].copy()
    # force an exact z == 0 for view 1 at voxel (1,1,1): zero that view's third row offset
    proj[0, 1, 2, :3] = np.array([0.0, 0.0, 1.0], dtype=np.float32)
    proj[0, 1, 2, 3] = 0.0
    coords[0, 1, 1, 1] = np.array([123.0, -45.0, 0.0], dtype=np.float32)
    run_unprojection_case("adversarial_v4c6", feats(1, 4, 6, 16, 16, 3, scale=3.0), proj, coords, 3)

    # 4. single view, and 8 views (MPI-INF-like ring), odd channel counts
    cams = ring_cameras(1, 5200.0, 1500.0, 1145.0, 1000.0, rng)
    P = feature_level_projections(cams, (150, 150, 850, 850), (64, 64), (10, 14))
    run_unprojection_case("single_view_c3", feats(2, 1, 3, 10, 14, 4), np.stack([P, P]),
                          np.stack([cuboid_coords((4, 4, 9), 2500.0)] * 2), 4)
    cams = ring_cameras(8, 4500.0, 1600.0, 1500.0, 2048.0, rng, jitter=0.08)
    P = feature_level_projections(cams, (300, 300, 1700, 1700), (96, 96), (24, 24))
    run_unprojection_case("eight_views_c7", feats(1, 8, 7, 24, 24, 5, scale=2.0), P[None],
                          cuboid_coords((8, 8, 8), 2500.0, theta=2.0)[None], 5)

    # 5. large-magnitude features (softmax saturation / max-subtraction path)
    cams = ring_cameras(4, 5000.0, 1500.0, 1145.0, 1000.0, rng)
    P = feature_level_projections(cams, (150, 150, 850, 850), (64, 64), (16, 16))
    run_unprojection_case("saturated_v4c4", feats(1, 4, 4, 16, 16, 6, scale=40.0), P[None],
                          cuboid_coords((6, 6, 6), 2500.0)[None], 6)

    # 6. BASELINE config[0] shape: 16^3 grid, 2 views, 64 ch, 56x56 maps, batch 1
    cams = ring_cameras(2, 5000.0, 1500.0, 1145.0, 1000.0, rng)
    P = feature_level_projections(cams, (150, 150, 850, 850), (224, 224), (56, 56))
    run_unprojection_case("config0_16cube", feats(1, 2, 64, 56, 56, 8), P[None],
                          cuboid_coords((16, 16, 16), 2500.0)[None], 8,
                          modes=("softmax", "max"), grad_modes=("softmax",))

    # 7. kernel-friendly shape: C = 16, z extent 32 (fast-path tiling), 4 views, theta != 0
    cams = ring_cameras(4, 5000.0, 1500.0, 1145.0, 1000.0, rng, jitter=0.03)
    P = feature_level_projections(cams, (150, 150, 850, 850), (128, 128), (32, 32))
    run_unprojection_case("tiles_v4c16", feats(2, 4, 16, 32, 32, 9), np.stack([P, P[::-1].copy()]),
                          np.stack([cuboid_coords((4, 8, 32), 2500.0, theta=0.4),
                                    cuboid_coords((4, 8, 32), 2500.0, center=(100.0, 50.0, -30.0), theta=5.1)]), 9)

    # 8. / 9. shapes whose tap windows fit the brick kernels' LDS pools (24x24 and 16x16 maps over the 2.5 m cuboid), so
    # that the LDS-staged forward and the LDS-accumulated backward are pinned by reference outputs, 4 and 8 views
    cams = ring_cameras(4, 5000.0, 1500.0, 1145.0, 1000.0, rng, jitter=0.02)
    P = feature_level_projections(cams, (150, 150, 850, 850), (96, 96), (24, 24))
    run_unprojection_case("bricks_v4c8", feats(2, 4, 8, 24, 24, 12), np.stack([P, P[::-1].copy()]),
                          np.stack([cuboid_coords((8, 8, 32), 2500.0, theta=0.0),
                                    cuboid_coords((8, 8, 32), 2500.0, center=(60.0, -40.0, 20.0), theta=2.3)]), 12)
    cams = ring_cameras(8, 5200.0, 1600.0, 1500.0, 2048.0, rng, jitter=0.02)
    P = feature_level_projections(cams, (300, 300, 1750, 1750), (64, 64), (16, 16))
    run_unprojection_case("bricks_v8c8", feats(1, 8, 8, 16, 16, 13), P[None],
                          cuboid_coords((4, 4, 64), 2500.0, theta=0.7)[None], 13)


# --------------------------------------------------------------------------- VolumeGenerator cases
def run_volgen_case(name, *, B, V, C_in, C_out, S, feat_hw, image_hw, kind, training, use_tri, seed):
    rng = np.random.default_rng(seed)
    cams = ring_cameras(V, 5000.0, 1500.0, 1145.0, 1000.0, rng, jitter=0.04)
    bbox = (150, 150, 850, 850)
    Ks, Rs, ts = [], [], []
    cameras = []
    for v in range(V):
        row = []
        for b in range(B):
            R, t, K = cams[v]
            cam = ref_mv.Camera(R, t + rng.normal(0, 5.0, 3), K)
            cam.update_after_crop(bbox)
            cam.update_after_resize((700, 700), (image_hw[1], image_hw[0]))
            row.append(cam)
            Ks.append(cam.K.copy()); Rs.append(cam.R.copy()); ts.append(cam.t.copy())
        cameras.append(row)
    keypoints = [rng.normal(0, 100.0, (17, 4)) for _ in range(B)]
    batch = dict(images=np.zeros((B, V, image_hw[0], image_hw[1], 3), dtype=np.uint8),
                 cameras=cameras, keypoints_3d=keypoints)
    torch.manual_seed(seed)
    gen = ref_agg.VolumeGenerator(volume_size=S, input_channels=C_in, output_channels=C_out,
                                  cuboid_side=2500.0, use_triangulation=use_tri, kind=kind, device="cpu")
    gen.train(training)
    features = torch.randn(B, V, C_in, *feat_hw)
    proj_org = torch.stack([torch.stack([torch.from_numpy(cameras[v][b].projection) for v in range(V)])
                            for b in range(B)]).float()
    captured = {}
    real = ref_agg.unprojection

    def spy(f, p, c, aggregation_method="softmax"):
        captured.update(features=f.detach().numpy().copy(), proj=p.numpy().copy(), coords=c.numpy().copy(),
                        method=aggregation_method)
        return real(f, p, c, aggregation_method=aggregation_method)

    ref_agg.unprojection = spy
    try:
        np.random.seed(seed)
        with torch.no_grad():
            vol = gen(features, proj_org, batch)
    finally:
        ref_agg.unprojection = real
    sd = gen.state_dict()
    rec = dict(K=np.stack(Ks).reshape(V, B, 3, 3), R=np.stack(Rs).reshape(V, B, 3, 3),
               t=np.stack(ts).reshape(V, B, 3, 1), keypoints=np.stack(keypoints),
               image_hw=np.array(image_hw), features_in=features.numpy(), proj_org=proj_org.numpy(),
               weight=sd["process_feature.0.weight"].numpy(), bias=sd["process_feature.0.bias"].numpy(),
               sd_keys=np.array(sorted(sd.keys())),
               features_conv=captured["features"], proj=captured["proj"], coords=captured["coords"],
               method=np.array(captured["method"]), volume=vol.numpy(),
               meta=np.array([B, V, C_in, C_out, S, int(training), int(use_tri), seed]), kind=np.array(kind))
    path = os.path.join(HERE, "volgen_%s.npz" % name)
    np.savez_compressed(path, **rec)
    print("wrote", path, "method=%s" % captured["method"])


def volgen_cases():
    run_volgen_case("eval_mpii", B=2, V=3, C_in=6, C_out=4, S=6, feat_hw=(12, 12), image_hw=(48, 48),
                    kind="mpii", training=False, use_tri=False, seed=11)
    run_volgen_case("train_mpii", B=3, V=2, C_in=5, C_out=3, S=5, feat_hw=(10, 14), image_hw=(40, 56),
                    kind="mpii", training=True, use_tri=False, seed=12)
    run_volgen_case("train_coco", B=2, V=4, C_in=4, C_out=4, S=4, feat_hw=(12, 12), image_hw=(48, 48),
                    kind="coco", training=True, use_tri=False, seed=13)
    run_volgen_case("eval_tri", B=2, V=4, C_in=4, C_out=2, S=4, feat_hw=(12, 12), image_hw=(48, 48),
                    kind="mpii", training=False, use_tri=True, seed=14)
    # the only combination in which the triangulated pivot reaches the output: training (theta != 0) + use_triangulation
    run_volgen_case("train_tri", B=3, V=4, C_in=4, C_out=4, S=6, feat_hw=(12, 12), image_hw=(48, 48),
                    kind="mpii", training=True, use_tri=True, seed=15)


# --------------------------------------------------------------------------- geometry helpers
def geometry_cases():
    rng = np.random.default_rng(21)
    rec = {}
    R, t, K = ring_cameras(1, 4000.0, 1200.0, 1100.0, 1000.0, rng)[0]
    cam = ref_mv.Camera(R, t, K, dist=[0.1, 0.2, 0.0, 0.0, 0.3])
    rec["cam_R"], rec["cam_t"], rec["cam_K"] = R, t, K
    rec["cam_P0"] = cam.projection
    rec["cam_ext0"] = cam.extrinsics
    cam.update_after_crop((100, 50, 900, 750))
    rec["cam_K_crop"] = cam.K.copy()
    cam.update_after_resize((700, 800), (96, 64))
    rec["cam_K_resize"] = cam.K.copy()
    rec["cam_P1"] = cam.projection
    rec["cam_dist"] = cam.dist
    pts = rng.normal(0, 300.0, (9, 3))
    rec["pts"] = pts
    rec["e2h_np"] = ref_mv.euclidean_to_homogeneous(pts)
    rec["e2h_t"] = ref_mv.euclidean_to_homogeneous(torch.from_numpy(pts)).numpy()
    hom = rng.normal(0, 1.0, (9, 4)) + 2.0
    rec["hom"] = hom
    rec["h2e_np"] = ref_mv.homogeneous_to_euclidean(hom)
    rec["h2e_t"] = ref_mv.homogeneous_to_euclidean(torch.from_numpy(hom)).numpy()
    P = rec["cam_P0"]
    rec["proj_np"] = ref_mv.project_3d_points_to_image_plane_without_distortion(P, pts)
    rec["proj_np_h"] = ref_mv.project_3d_points_to_image_plane_without_distortion(P, pts, convert_back_to_euclidean=False)
    rec["proj_t"] = ref_mv.project_3d_points_to_image_plane_without_distortion(
        torch.from_numpy(P), torch.from_numpy(pts)).numpy()
    # DLT triangulation (numpy + torch), 4 views of one point
    cams = ring_cameras(4, 4500.0, 1500.0, 1145.0, 1000.0, rng, jitter=0.05)
    Ps = np.stack([ref_mv.Camera(*c).projection for c in cams])
    X = np.array([120.0, -80.0, 40.0])
    uv = np.stack([ref_mv.project_3d_points_to_image_plane_without_distortion(p, X[None])[0] for p in Ps])
    uv_noisy = uv + rng.normal(0, 0.5, uv.shape)
    rec["tri_P"], rec["tri_uv"] = Ps, uv_noisy
    rec["tri_np"] = ref_mv.triangulate_point_from_multiple_views_linear(Ps, uv_noisy)
    rec["tri_t"] = ref_mv.triangulate_point_from_multiple_views_linear_torch(
        torch.from_numpy(Ps).float(), torch.from_numpy(uv_noisy).float()).numpy()
    conf = np.array([1.0, 0.5, 0.8, 0.2], dtype=np.float32)
    rec["tri_conf"] = conf
    rec["tri_t_conf"] = ref_mv.triangulate_point_from_multiple_views_linear_torch(
        torch.from_numpy(Ps).float(), torch.from_numpy(uv_noisy).float(), torch.from_numpy(conf)).numpy()
    # rotations
    thetas = np.array([0.0, 0.3, 2.0, 5.5])
    rec["rot_thetas"] = thetas
    rec["rot_z"] = np.stack([ref_vol.get_rotation_matrix([0, 0, 1], th) for th in thetas])
    rec["rot_y"] = np.stack([ref_vol.get_rotation_matrix([0, 1, 0], th) for th in thetas])
    rec["rot_arb"] = np.stack([ref_vol.get_rotation_matrix([1, 2, -0.5], th) for th in thetas])
    cv = torch.from_numpy(rng.normal(0, 500.0, (3, 4, 5, 3)).astype(np.float32))
    rec["rcv_in"] = cv.numpy()
    rec["rcv_out"] = ref_vol.rotate_coord_volume(cv, 1.234, [0, 0, 1]).numpy()
    cub = ref_vol.Cuboid3D(np.array([-1.0, -2.0, -3.0]), np.array([2.0, 4.0, 6.0]))
    rec["cub_pos"], rec["cub_sides"] = cub.position, cub.sides
    path = os.path.join(HERE, "geometry.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path)


if __name__ == "__main__":
    unprojection_cases()
    if not ONLY:
        volgen_cases()
        geometry_cases()
