"""The reference's CPU algorithm restated with the same ATen ops -- the timed CPU baseline ("port").

TEST INFRASTRUCTURE ONLY.  Structurally the loop of models/aggregation.py:28-83: one
F.grid_sample(bilinear, zeros, align_corners=True) per (b, v) into a (V,C,X,Y,Z) stack, masked,
then the cross-view aggregate -- i.e. what a user of the reference runs on host cores today.
Written from the algorithm description in SURVEY.md section 8(a); validated against the goldens
in tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F


def unprojection_cpu_loop(features, proj_matricies, coord_volumes, aggregation_method="softmax"):
    B, V, C, Hf, Wf = features.shape
    vol = tuple(coord_volumes.shape[1:4])
    result = features.new_zeros((B, C) + vol, dtype=torch.float32)
    scale = features.new_tensor([float(Hf), float(Wf)])          # quirk Q1: x / Hf, y / Wf
    for b in range(B):
        pts = coord_volumes[b].reshape(-1, 3)
        hom = torch.cat([pts, pts.new_ones(pts.shape[0], 1)], dim=1)
        stack = features.new_zeros((V, C) + vol)
        for v in range(V):
            pr = hom @ proj_matricies[b, v].t()
            behind = pr[:, 2] <= 0
            depth = torch.where(pr[:, 2] == 0, torch.ones_like(pr[:, 2]), pr[:, 2])
            grid = 2.0 * (pr[:, :2] / depth[:, None] / scale - 0.5)
            sampled = F.grid_sample(features[b, v][None], grid[None, :, None, :], mode="bilinear",
                                    padding_mode="zeros", align_corners=True)
            sampled = sampled.reshape(C, -1).masked_fill(behind[None], 0.0)
            stack[v] = sampled.reshape((C,) + vol)
        if aggregation_method == "sum":
            result[b] = stack.sum(0)
        elif aggregation_method == "mean":
            result[b] = stack.mean(0)
        elif aggregation_method == "max":
            result[b] = stack.max(0)[0]
        elif aggregation_method == "softmax":
            weights = torch.softmax(stack.reshape(V, -1), dim=0).reshape(stack.shape)
            result[b] = (stack * weights).sum(0)
        else:
            raise ValueError("Unknown aggregation_method: {}".format(aggregation_method))
    return result
