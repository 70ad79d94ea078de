"""Volume helpers with the interface of the reference's utils/volumetric.py (the parts on the path).

`Cuboid3D` (:44-47), `get_rotation_matrix` (:87-99) and `rotate_coord_volume` (:102-114) are what
VolumeGenerator uses; the cv2 drawing classes (Point3D / Line3D / Cuboid3D.render, :8-41, :49-84)
are visualisation and out of scope.
"""
import numpy as np
import torch


class Cuboid3D:
    """Axis-aligned box: `position` is the min corner, `sides` the edge lengths (numpy, mm)."""

    def __init__(self, position, sides):
        self.position = position
        self.sides = sides


def get_rotation_matrix(axis, theta):
    """Counter-clockwise rotation by `theta` rad about `axis`, float64 (3,3)  (utils/volumetric.py:87-99).

    Unit quaternion (w, v) with w = cos(theta/2), v = -axis/|axis| * sin(theta/2), expanded as
    R = (w^2 - v.v) I + 2 v v^T + 2 w S(v), the same matrix the reference spells out entry by entry.
    """
    axis = np.asarray(axis, dtype=np.float64)
    w = np.cos(theta / 2.0)
    v = -(axis / np.sqrt(axis @ axis)) * np.sin(theta / 2.0)
    skew = np.array([[0.0, v[2], -v[1]], [-v[2], 0.0, v[0]], [v[1], -v[0], 0.0]])
    return (w * w - v @ v) * np.eye(3) + 2.0 * np.outer(v, v) + 2.0 * w * skew


def get_rotation_matrices(axis, thetas):
    """get_rotation_matrix for an array of angles at once: (B,3,3) float64, entry for entry the same expressions."""
    axis = np.asarray(axis, dtype=np.float64)
    thetas = np.asarray(thetas, dtype=np.float64)
    w = np.cos(thetas / 2.0)                                            # (B,)
    v = -(axis / np.sqrt(axis @ axis))[None, :] * np.sin(thetas / 2.0)[:, None]     # (B,3)
    zero = np.zeros_like(w)
    skew = np.stack([np.stack([zero, v[:, 2], -v[:, 1]], -1), np.stack([-v[:, 2], zero, v[:, 0]], -1),
                     np.stack([v[:, 1], -v[:, 0], zero], -1)], -2)
    vv = np.einsum("bi,bi->b", v, v)
    return (w * w - vv)[:, None, None] * np.eye(3)[None] + 2.0 * v[:, :, None] * v[:, None, :] + 2.0 * w[:, None, None] * skew


def rotate_coord_volume(coord_volume, theta, axis):
    """Rotate every point of a (..., 3) coordinate volume about the origin; fp32 rotation like the reference."""
    rot = torch.from_numpy(get_rotation_matrix(axis, theta)).type(torch.float).to(coord_volume.device)
    flat = coord_volume.reshape(-1, 3)
    return rot.mm(flat.t()).t().reshape(coord_volume.shape)
