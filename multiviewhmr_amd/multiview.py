"""Multi-view geometry helpers with the interface of the reference's utils/multiview.py.

Same names, argument meaning, dtypes and error behaviour (utils/multiview.py:5-168), written for
this package: the hot loop itself never calls these (the projection lives inside the HIP kernel,
csrc/device_common.h::make_taps); they serve the caller side (`VolumeGenerator`) and user code.
"""
import numpy as np
import torch

_TYPE_MSG = "Works only with numpy arrays and PyTorch tensors."   # utils/multiview.py:69,86,110


class Camera:
    """Pin-hole camera, float64 numpy K (3,3), R (3,3), t (3,1)  (utils/multiview.py:5-52)."""

    def __init__(self, R, t, K, dist=None, name=""):
        self.R = np.array(R, copy=True)
        assert self.R.shape == (3, 3)
        t = np.array(t, copy=True)
        assert t.size == 3
        self.t = t.reshape(3, 1)
        self.K = np.array(K, copy=True)
        assert self.K.shape == (3, 3)
        self.dist = None if dist is None else np.array(dist, copy=True).flatten()
        self.name = name

    def update_after_crop(self, bbox):
        """bbox = (left, upper, right, lower): the principal point moves with the crop origin (:23-31)."""
        left, upper = bbox[0], bbox[1]
        self.K[0, 2] = self.K[0, 2] - left
        self.K[1, 2] = self.K[1, 2] - upper

    def update_after_resize(self, image_shape, new_image_shape):
        """image_shape is (height, width); new_image_shape is read as (WIDTH, HEIGHT) -- quirk Q3 of the
        reference (:34-35), kept on purpose: VolumeGenerator passes (Hf, Wf) here (aggregation.py:130)."""
        height, width = image_shape
        new_width, new_height = new_image_shape
        sx, sy = new_width / width, new_height / height
        self.K[0, 0], self.K[1, 1] = self.K[0, 0] * sx, self.K[1, 1] * sy
        self.K[0, 2], self.K[1, 2] = self.K[0, 2] * sx, self.K[1, 2] * sy

    @property
    def extrinsics(self):
        return np.hstack([self.R, self.t])

    @property
    def projection(self):
        return self.K.dot(self.extrinsics)


def euclidean_to_homogeneous(points):
    """(N, M) -> (N, M+1) by appending ones (utils/multiview.py:55-69)."""
    if isinstance(points, np.ndarray):
        return np.hstack([points, np.ones((len(points), 1))])
    if torch.is_tensor(points):
        ones = torch.ones((points.shape[0], 1), dtype=points.dtype, device=points.device)
        return torch.cat([points, ones], dim=1)
    raise TypeError(_TYPE_MSG)


def homogeneous_to_euclidean(points):
    """(N, M+1) -> (N, M): divide by the last coordinate; also accepts a single (M+1,) vector like the
    reference's transpose-based form does (utils/multiview.py:72-86)."""
    if isinstance(points, np.ndarray):
        return (points.T[:-1] / points.T[-1]).T
    if torch.is_tensor(points):
        pt = points.transpose(1, 0)
        return (pt[:-1] / pt[-1]).transpose(1, 0)
    raise TypeError(_TYPE_MSG)


def project_3d_points_to_image_plane_without_distortion(proj_matrix, points_3d, convert_back_to_euclidean=True):
    """[X, 1] @ P^T, optionally de-homogenised (utils/multiview.py:89-110)."""
    both_np = isinstance(proj_matrix, np.ndarray) and isinstance(points_3d, np.ndarray)
    both_t = torch.is_tensor(proj_matrix) and torch.is_tensor(points_3d)
    if not (both_np or both_t):
        raise TypeError(_TYPE_MSG)
    result = euclidean_to_homogeneous(points_3d) @ (proj_matrix.T if both_np else proj_matrix.t())
    return homogeneous_to_euclidean(result) if convert_back_to_euclidean else result


def triangulate_point_from_multiple_views_linear(proj_matricies, points):
    """DLT triangulation of one point, numpy (utils/multiview.py:113-138)."""
    assert len(proj_matricies) == len(points)
    rows = []
    for P, uv in zip(proj_matricies, points):
        rows.append(uv[0] * P[2, :] - P[0, :])
        rows.append(uv[1] * P[2, :] - P[1, :])
    _, _, vh = np.linalg.svd(np.asarray(rows, dtype=np.float64), full_matrices=False)
    return homogeneous_to_euclidean(vh[3, :])


def triangulate_point_from_multiple_views_linear_torch(proj_matricies, points, confidences=None):
    """DLT triangulation of one point, torch (utils/multiview.py:141-168); confidences weight the rows."""
    assert len(proj_matricies) == len(points)
    n_views = len(proj_matricies)
    if confidences is None:
        confidences = torch.ones(n_views, dtype=torch.float32, device=points.device)
    A = proj_matricies[:, 2:3].expand(n_views, 2, 4) * points.view(n_views, 2, 1) - proj_matricies[:, :2]
    A = A * confidences.view(-1, 1, 1)
    _, _, v = torch.svd(A.reshape(-1, 4))
    return homogeneous_to_euclidean((-v[:, 3]).unsqueeze(0))[0]


def triangulate_points_from_multiple_views_linear_batch(proj_matricies, points, confidences=None):
    """The same DLT for a whole batch at once: proj_matricies (B, V, 3, 4), points (V, 2) shared by the samples -> (B, 3),
    on proj_matricies.device with no host synchronisation (HIP tensors: mvhmr_triangulate_dlt; CPU tensors -- tests without a GPU --
    the batched float64 SVD below).  The caller (VolumeGenerator with use_triangulation, reference
    aggregation.py:174-177) triangulates the image centre of every sample; the reference does it sample by sample with a
    device SVD and a .cpu() each.  One batched float64 SVD of the (B, 2V, 4) system here; the right singular vector of the
    smallest singular value is the homogeneous point (its sign cancels in the dehomogenisation)."""
    B, V = proj_matricies.shape[:2]
    if proj_matricies.is_cuda:
        # the library's SVD-free kernel: one thread per sample, 4 x 4 normal matrix, float64 Jacobi rotations
        import ctypes
        from . import _capi
        L = _capi.lib()
        P32 = proj_matricies.detach().to(torch.float32).contiguous()
        pts32 = points.detach().to(device=P32.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, 3, dtype=torch.float32, device=P32.device)
        per_sample = 1 if pts32.dim() == 3 else 0
        with torch.cuda.device(P32.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(P32.device).cuda_stream)
            if confidences is None:
                _capi.check(L.mvhmr_triangulate_dlt(ctypes.c_void_p(P32.data_ptr()), ctypes.c_void_p(pts32.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                    B, V, per_sample, stream))
            else:                                                             # A *= confidences (utils/multiview.py:156-161), (V,) or (B, V)
                c32 = confidences.detach().to(device=P32.device, dtype=torch.float32).contiguous()
                _capi.check(L.mvhmr_triangulate_dlt_weighted(ctypes.c_void_p(P32.data_ptr()), ctypes.c_void_p(pts32.data_ptr()), ctypes.c_void_p(c32.data_ptr()),
                                                             ctypes.c_void_p(out.data_ptr()), B, V, per_sample, 1 if c32.dim() == 2 else 0, stream))
        return out
    P = proj_matricies.to(torch.float64)
    pts = points.to(device=P.device, dtype=torch.float64)
    A = P[:, :, 2:3].expand(B, V, 2, 4) * (pts.view(B, V, 2, 1) if pts.dim() == 3 else pts.view(1, V, 2, 1)) - P[:, :, :2]
    if confidences is not None:
        c = confidences.to(device=P.device, dtype=torch.float64)
        A = A * (c.view(B, V, 1, 1) if c.dim() == 2 else c.view(1, V, 1, 1))
    _, _, vh = torch.linalg.svd(A.reshape(B, 2 * V, 4), full_matrices=False)
    h = vh[:, 3, :]
    return (h[:, :3] / h[:, 3:4]).to(torch.float32)
