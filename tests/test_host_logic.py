"""CPU tests of the host side that mirrors the reference interface: geometry helpers against goldens
recorded from the reference, VolumeGenerator's caller-side logic, argument checking and error types."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from multiviewhmr_amd import aggregation, multiview, volumetric


@pytest.fixture(scope="module")
def geo():
    import os, conftest
    return np.load(os.path.join(conftest.GOLDEN, "geometry.npz"))


def test_camera_bookkeeping(geo):
    cam = multiview.Camera(geo["cam_R"], geo["cam_t"], geo["cam_K"], dist=[0.1, 0.2, 0.0, 0.0, 0.3])
    np.testing.assert_array_equal(cam.projection, geo["cam_P0"])
    np.testing.assert_array_equal(cam.extrinsics, geo["cam_ext0"])
    cam.update_after_crop((100, 50, 900, 750))
    np.testing.assert_array_equal(cam.K, geo["cam_K_crop"])
    cam.update_after_resize((700, 800), (96, 64))          # quirk Q3: second argument is (W, H)
    np.testing.assert_array_equal(cam.K, geo["cam_K_resize"])
    np.testing.assert_array_equal(cam.projection, geo["cam_P1"])
    np.testing.assert_array_equal(cam.dist, geo["cam_dist"])
    assert cam.t.shape == (3, 1)
    with pytest.raises(AssertionError):
        multiview.Camera(np.eye(4), np.zeros(3), np.eye(3))


def test_camera_copies_its_inputs(geo):
    K = geo["cam_K"].copy()
    cam = multiview.Camera(geo["cam_R"], geo["cam_t"], K)
    cam.update_after_crop((10, 20, 30, 40))
    np.testing.assert_array_equal(K, geo["cam_K"])


def test_homogeneous_helpers(geo):
    np.testing.assert_array_equal(multiview.euclidean_to_homogeneous(geo["pts"]), geo["e2h_np"])
    np.testing.assert_array_equal(multiview.euclidean_to_homogeneous(torch.from_numpy(geo["pts"])).numpy(), geo["e2h_t"])
    np.testing.assert_array_equal(multiview.homogeneous_to_euclidean(geo["hom"]), geo["h2e_np"])
    np.testing.assert_array_equal(multiview.homogeneous_to_euclidean(torch.from_numpy(geo["hom"])).numpy(), geo["h2e_t"])
    for fn in (multiview.euclidean_to_homogeneous, multiview.homogeneous_to_euclidean):
        with pytest.raises(TypeError, match="Works only with numpy arrays and PyTorch tensors"):
            fn([[1.0, 2.0, 3.0]])


def test_projection_helper(geo):
    P, pts = geo["cam_P0"], geo["pts"]
    f = multiview.project_3d_points_to_image_plane_without_distortion
    np.testing.assert_array_equal(f(P, pts), geo["proj_np"])
    np.testing.assert_array_equal(f(P, pts, convert_back_to_euclidean=False), geo["proj_np_h"])
    np.testing.assert_array_equal(f(torch.from_numpy(P), torch.from_numpy(pts)).numpy(), geo["proj_t"])
    with pytest.raises(TypeError):
        f(P, torch.from_numpy(pts))                          # mixed numpy / torch, multiview.py:110


def test_batched_triangulation_matches_the_per_sample_form():
    """VolumeGenerator(use_triangulation=True) triangulates every sample's image centre in one batched call"""
    rng = np.random.default_rng(4)
    P = []
    for b in range(5):
        Pb = []
        for v in range(4):
            az = 2 * np.pi * v / 4 + 0.2 * b
            eye = np.array([4500 * np.cos(az), 4500 * np.sin(az), 1400.0 + 50 * b])
            fwd = -eye / np.linalg.norm(eye); right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
            R = np.stack([right, np.cross(fwd, right), fwd])
            Pb.append(multiview.Camera(R, -R @ eye, [[1100.0, 0, 190], [0, 1100.0, 200], [0, 0, 1]]).projection)
        P.append(Pb)
    P = torch.from_numpy(np.asarray(P, dtype=np.float32))
    uv = torch.tensor([[192.0, 192.0]]).expand(4, 2)
    got = multiview.triangulate_points_from_multiple_views_linear_batch(P, uv).numpy()
    for b in range(5):
        ref = multiview.triangulate_point_from_multiple_views_linear(P[b].numpy().astype(np.float64), uv.numpy().astype(np.float64))
        np.testing.assert_allclose(got[b], ref, rtol=0, atol=2e-2)        # mm; float32 inputs, float64 solves


def test_triangulation(geo):
    np.testing.assert_allclose(multiview.triangulate_point_from_multiple_views_linear(geo["tri_P"], geo["tri_uv"]),
                               geo["tri_np"], rtol=1e-9, atol=1e-9)
    P, uv = torch.from_numpy(geo["tri_P"]).float(), torch.from_numpy(geo["tri_uv"]).float()
    np.testing.assert_allclose(multiview.triangulate_point_from_multiple_views_linear_torch(P, uv).numpy(), geo["tri_t"],
                               rtol=1e-4, atol=1e-2)
    got = multiview.triangulate_point_from_multiple_views_linear_torch(P, uv, torch.from_numpy(geo["tri_conf"])).numpy()
    np.testing.assert_allclose(got, geo["tri_t_conf"], rtol=1e-4, atol=1e-2)


def test_rotation_helpers(geo):
    for key, axis in (("rot_z", [0, 0, 1]), ("rot_y", [0, 1, 0]), ("rot_arb", [1, 2, -0.5])):
        got = np.stack([volumetric.get_rotation_matrix(axis, th) for th in geo["rot_thetas"]])
        np.testing.assert_allclose(got, geo[key], rtol=0, atol=1e-15)
    out = volumetric.rotate_coord_volume(torch.from_numpy(geo["rcv_in"]), 1.234, [0, 0, 1]).numpy()
    np.testing.assert_allclose(out, geo["rcv_out"], rtol=0, atol=1e-4)
    cub = volumetric.Cuboid3D(geo["cub_pos"], geo["cub_sides"])
    np.testing.assert_array_equal(cub.position, geo["cub_pos"])
    np.testing.assert_array_equal(cub.sides, geo["cub_sides"])


# ------------------------------------------------------------------ VolumeGenerator caller side (no kernel)
def _rebuild(d):
    B, V, C_in, C_out, S, training, use_tri, seed = (int(x) for x in d["meta"])
    cams = [[multiview.Camera(d["R"][v, b], d["t"][v, b], d["K"][v, b]) for b in range(B)] for v in range(V)]
    batch = dict(images=np.zeros((B, V, int(d["image_hw"][0]), int(d["image_hw"][1]), 3), dtype=np.uint8),
                 cameras=cams, keypoints_3d=[k for k in d["keypoints"]])
    gen = aggregation.VolumeGenerator(volume_size=S, input_channels=C_in, output_channels=C_out, cuboid_side=2500.0,
                                      use_triangulation=bool(use_tri), kind=str(d["kind"]), device="cpu")
    gen.load_state_dict({"process_feature.0.weight": torch.from_numpy(d["weight"]),
                         "process_feature.0.bias": torch.from_numpy(d["bias"])})
    gen.train(bool(training))
    return gen, batch, seed


def _host_coords(gen, rots, centers):
    """numpy rendition of what mvhmr_build_coord_volumes computes on the device (float64, for tolerance checks)."""
    S, cub = gen.volume_size, gen.cuboid()
    idx = np.stack(np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij"), -1).astype(np.float64)
    grid = cub.position + cub.sides / (S - 1) * idx
    out = []
    for R, c in zip(rots.numpy().astype(np.float64), centers.numpy().astype(np.float64)):
        out.append((grid - c) @ R.T + c)
    return np.stack(out)


@pytest.mark.parametrize("case", golden_cases("volgen"))
def test_volume_generator_geometry_matches_reference(case):
    d = load_golden("volgen", case)
    gen, batch, seed = _rebuild(d)
    assert sorted(gen.state_dict().keys()) == list(d["sd_keys"])          # checkpoints load unchanged
    feat_hw = tuple(d["features_in"].shape[-2:])
    proj = aggregation.feature_level_projections(batch["cameras"], tuple(batch["images"].shape[2:-1]), feat_hw)
    np.testing.assert_array_equal(proj, d["proj"])                        # same float64 arithmetic, Q3 included
    K_before = [[c.K.copy() for c in row] for row in batch["cameras"]]
    np.random.seed(seed)
    rots, centers = gen.volume_pose(batch, torch.from_numpy(d["proj_org"]), tuple(batch["images"].shape[2:-1]))
    for row, rowK in zip(batch["cameras"], K_before):                     # caller's cameras untouched
        for c, K in zip(row, rowK):
            np.testing.assert_array_equal(c.K, K)
    coords = _host_coords(gen, rots, centers)
    tol = 5e-2 if int(d["meta"][6]) else 2e-3                             # triangulated pivot goes through an SVD
    np.testing.assert_allclose(coords, d["coords"], rtol=0, atol=tol)
    # 1x1 conv (stays on PyTorch): same weights -> same post-conv features as the reference handed to unprojection
    with torch.no_grad():
        f = torch.from_numpy(d["features_in"])
        conv = gen.process_feature(f.view(-1, *f.shape[2:])).view(f.shape[0], f.shape[1], -1, *f.shape[3:])
    np.testing.assert_allclose(conv.numpy(), d["features_conv"], rtol=0, atol=1e-5)
    assert str(d["method"]) == "softmax"                                  # quirk Q5


def test_training_rotation_uses_global_numpy_stream():
    d = load_golden("volgen", "train_mpii")
    gen, batch, seed = _rebuild(d)
    np.random.seed(seed)
    r1, _ = gen.volume_pose(batch, torch.from_numpy(d["proj_org"]), (40, 56))
    np.random.seed(seed)
    expected = np.stack([volumetric.get_rotation_matrix([0, 0, 1], np.random.uniform(0.0, 2 * np.pi))
                         for _ in range(r1.shape[0])]).astype(np.float32)
    np.testing.assert_array_equal(r1.numpy(), expected)
    gen.eval()
    r2, _ = gen.volume_pose(batch, torch.from_numpy(d["proj_org"]), (40, 56))
    np.testing.assert_array_equal(r2.numpy(), np.broadcast_to(np.eye(3, dtype=np.float32), r2.shape))


class _Node(dict):
    __getattr__ = dict.__getitem__


def test_build_volume_generator_wiring_and_q5():
    cfg = _Node(MODEL=_Node(BACKBONE=_Node(DECONV_FILTERS=[256, 256, 64], DECONV_LAYERS=3),
                            AGGREGATION=_Node(VOLUME_SIZE=16, OUTPUT_CHANNELS=8, CUBOID_SIDE=2000.0, USE_TRIANGULATION=False,
                                              METHOD="mean")),
                DATASET=_Node(KIND="coco", TYPE="human36m"))
    torch_default = torch.get_default_dtype()
    import unittest.mock as mock
    with mock.patch.object(aggregation.VolumeGenerator, "to", lambda self, *a, **k: self):   # no HIP device here
        gen = aggregation.build_volume_generator(cfg)
    assert gen.process_feature[0].in_channels == 64 and gen.process_feature[0].out_channels == 8
    assert gen.volume_size == 16 and gen.cuboid_side == 2000.0 and gen.kind == "coco"
    assert gen.aggregation_method == "softmax"            # METHOD='mean' is swallowed by **kwargs (Q5)
    cfg.MODEL.BACKBONE.DECONV_LAYERS = 0
    with mock.patch.object(aggregation.VolumeGenerator, "to", lambda self, *a, **k: self):
        assert aggregation.build_volume_generator(cfg).process_feature[0].in_channels == 2048
    assert torch.get_default_dtype() == torch_default


# ------------------------------------------------------------------ argument checking of unprojection
def _tiny():
    d = load_golden("unproj", "tiny_b2v2c4")
    return torch.from_numpy(d["features"]), torch.from_numpy(d["proj"]), torch.from_numpy(d["coords"])


def test_unprojection_error_types():
    f, p, c = _tiny()
    with pytest.raises(ValueError, match="Unknown aggregation_method: median"):
        aggregation.unprojection(f, p, c, aggregation_method="median")
    with pytest.raises(TypeError, match="Works only with numpy arrays and PyTorch tensors"):
        aggregation.unprojection(f, p.numpy(), c)
    with pytest.raises(RuntimeError, match="proj_matricies must be"):
        aggregation.unprojection(f, p[:, :1], c)
    with pytest.raises(RuntimeError, match="coord_volumes must be"):
        aggregation.unprojection(f, p, c[..., :2])


def test_unprojection_has_no_cpu_fallback():
    f, p, c = _tiny()
    with pytest.raises(RuntimeError, match="no CPU path"):
        aggregation.unprojection(f, p, c)


def test_batched_rotation_matrices_and_sized_theta_draws_match_the_per_sample_loop():
    """VolumeGenerator.volume_pose draws all thetas with one sized call and builds the rotations vectorised: both must equal
    the reference's per-sample loop (aggregation.py:163-167, volumetric.py:87-99), bit for bit"""
    np.random.seed(123)
    seq = np.array([np.random.uniform(0.0, 2 * np.pi) for _ in range(7)])
    np.random.seed(123)
    assert np.array_equal(seq, np.random.uniform(0.0, 2 * np.pi, size=7))
    for axis in ([0, 0, 1], [0, 1, 0]):                                  # the two axes VolumeGenerator uses ('mpii' / 'coco'): bit-exact
        batched = volumetric.get_rotation_matrices(axis, seq)
        for b, th in enumerate(seq):
            assert np.array_equal(batched[b], volumetric.get_rotation_matrix(axis, th))
    batched = volumetric.get_rotation_matrices([1.0, 2.0, -0.5], seq)    # any other axis: same expressions, summation order of v.v may differ
    for b, th in enumerate(seq):
        assert np.abs(batched[b] - volumetric.get_rotation_matrix([1.0, 2.0, -0.5], th)).max() <= 4e-16
    assert np.array_equal(volumetric.get_rotation_matrix([0, 0, 1], 0.0), np.eye(3))


def test_device_projection_recipe_matches_the_host_one_on_cpu_tensors():
    """feature_level_projections_device is plain torch float64 arithmetic: on CPU tensors it must reproduce the numpy path (and
    hence the reference, test above) bit for bit -- the GPU runs the same IEEE operations"""
    rng = np.random.default_rng(5)
    V, B = 3, 4
    cams = [[multiview.Camera(np.linalg.qr(rng.standard_normal((3, 3)))[0], rng.standard_normal((3, 1)) * 1000,
                              np.array([[1100.0 + b, 0, 500 + v], [0, 1090.0 - v, 510 + b], [0, 0, 1.0]])) for b in range(B)] for v in range(V)]
    host = aggregation.feature_level_projections(cams, (384, 320), (96, 80))
    dev = aggregation.feature_level_projections_device(aggregation.pack_cameras(cams, "cpu"), (384, 320), (96, 80))
    assert dev.dtype == torch.float32 and np.array_equal(dev.numpy(), host)


def test_ops_are_registered_with_torch_library_and_trace_without_a_gpu():
    """mvhmr::unprojection / ::unprojection_cuboid (+ _backward): shape functions and the autograd formula are registered, so FakeTensor
    tracing (torch.compile / AOT autograd) sees one node with a known output and a known backward -- no HIP device needed for that"""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from multiviewhmr_amd import aggregation, _capi  # noqa: F401  (registers the ops)
    f = torch.empty(2, 4, 8, 24, 24, device="meta")
    out = torch.ops.mvhmr.unprojection(f, torch.empty(2, 4, 3, 4, device="meta"), torch.empty(2, 8, 8, 32, 3, device="meta"), 0, _capi.BF16, 0)
    assert tuple(out.shape) == (2, 8, 8, 8, 32) and out.dtype == torch.bfloat16
    with FakeTensorMode():
        f = torch.empty(2, 4, 8, 24, 24, requires_grad=True)
        p = torch.empty(2, 4, 3, 4)
        o = torch.ops.mvhmr.unprojection(f, p, torch.empty(2, 8, 8, 32, 3), 0, _capi.F32, 0)
        o.sum().backward()
        assert tuple(o.shape) == (2, 8, 8, 8, 32) and tuple(f.grad.shape) == tuple(f.shape)
        f2 = torch.empty(2, 4, 8, 24, 24, requires_grad=True)
        o2 = torch.ops.mvhmr.unprojection_cuboid(f2, p, torch.empty(2, 3, 3), torch.empty(2, 3), [0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [4, 8, 32], 1, _capi.F32, 0)
        o2.sum().backward()
        assert tuple(o2.shape) == (2, 8, 4, 8, 32) and tuple(f2.grad.shape) == tuple(f2.shape)
