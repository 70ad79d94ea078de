"""numpy restatement of the reference un-projection (vectorised; small and medium cases).

TEST INFRASTRUCTURE ONLY -- see oracle/unproject_oracle.c for the reference file:line map; this
file restates the same steps a second, independent way (array ops instead of scalar loops) so
that the two oracles cross-check each other as well as the goldens.
"""
import numpy as np

F32 = np.float32


def tap_table(proj_bv, pts, Hf, Wf):
    """One (b, v): pts (N,3) -> flat tap offsets (4,N) int64, weights (4,N) f32 (0 where out of range
    or z <= 0).  Follows models/aggregation.py:38-58 + ATen grid_sampler_2d (bilinear/zeros/align_corners)."""
    P = proj_bv.astype(F32)
    hom = np.concatenate([pts.astype(F32), np.ones((len(pts), 1), F32)], axis=1)      # multiview.py:67
    a, b, z = (hom * P[0]).sum(1, dtype=F32), (hom * P[1]).sum(1, dtype=F32), (hom * P[2]).sum(1, dtype=F32)
    invalid = z <= 0                                                                    # aggregation.py:42
    zs = np.where(z == 0, F32(1), z)                                                    # aggregation.py:44
    with np.errstate(all="ignore"):
        u, v = a / zs, b / zs                                                           # multiview.py:84
        gx = F32(2) * (u / F32(Hf) - F32(0.5))                                          # aggregation.py:49 (Q1)
        gy = F32(2) * (v / F32(Wf) - F32(0.5))                                          # aggregation.py:50 (Q1)
        ix = ((gx + F32(1)) / F32(2)) * F32(Wf - 1)                                     # align_corners=True
        iy = ((gy + F32(1)) / F32(2)) * F32(Hf - 1)
        inside = (ix > -1) & (ix < Wf) & (iy > -1) & (iy < Hf) & ~invalid
    ix = np.where(inside, ix, F32(0)); iy = np.where(inside, iy, F32(0))
    x0f, y0f = np.floor(ix), np.floor(iy)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    wx1, wy1 = ix - x0f, iy - y0f
    wx0, wy0 = (x0f + F32(1)) - ix, (y0f + F32(1)) - iy
    xs = np.stack([x0, x0 + 1, x0, x0 + 1]); ys = np.stack([y0, y0, y0 + 1, y0 + 1])
    ws = np.stack([wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1]).astype(F32)
    ok = (xs >= 0) & (xs < Wf) & (ys >= 0) & (ys < Hf) & inside[None]
    off = np.where(ok, ys * Wf + xs, 0)
    return off, np.where(ok, ws, F32(0)), ok


def per_view_samples(features, proj, coords):
    """-> s (B,V,C,N) f32 : the reference's volume_batch_to_aggregate, never formed by the product."""
    B, V, C, Hf, Wf = features.shape
    N = int(np.prod(coords.shape[1:4]))
    s = np.zeros((B, V, C, N), F32)
    tables = {}
    for b in range(B):
        pts = coords[b].reshape(-1, 3)
        for v in range(V):
            off, w, ok = tap_table(proj[b, v], pts, Hf, Wf)
            tables[b, v] = (off, w, ok)
            planes = features[b, v].reshape(C, Hf * Wf).astype(F32)
            acc = np.zeros((C, N), F32)
            for k in range(4):                                   # nw, ne, sw, se in ATen's order
                acc += planes[:, off[k]] * w[k][None]
            s[b, v] = acc
    return s, tables


def aggregate(s, method):
    V = s.shape[1]
    if method == "sum":
        return s.sum(1, dtype=F32)
    if method == "mean":
        return s.sum(1, dtype=F32) / F32(V)
    if method == "max":
        return s.max(1)
    if method == "softmax":
        e = np.exp(s - s.max(1, keepdims=True), dtype=F32)
        p = e / e.sum(1, keepdims=True, dtype=F32)
        return (s * p).sum(1, dtype=F32)
    raise ValueError("Unknown aggregation_method: {}".format(method))


def forward(features, proj, coords, method="softmax"):
    s, _ = per_view_samples(features, proj, coords)
    B, _, C, _ = s.shape
    return aggregate(s, method).reshape((B, C) + tuple(coords.shape[1:4]))


def backward(grad_out, features, proj, coords, method="softmax"):
    B, V, C, Hf, Wf = features.shape
    s, tables = per_view_samples(features, proj, coords)
    g = grad_out.reshape(B, 1, C, -1).astype(F32)
    if method == "sum":
        ds = np.broadcast_to(g, s.shape)
    elif method == "mean":
        ds = np.broadcast_to(g / F32(V), s.shape)
    elif method == "max":
        ds = g * (np.arange(V)[None, :, None, None] == s.argmax(1)[:, None])
    elif method == "softmax":
        e = np.exp(s - s.max(1, keepdims=True), dtype=F32)
        p = e / e.sum(1, keepdims=True, dtype=F32)
        o = (s * p).sum(1, keepdims=True, dtype=F32)
        ds = g * p * (F32(1) + s - o)
    else:
        raise ValueError("Unknown aggregation_method: {}".format(method))
    gf = np.zeros((B, V, C, Hf * Wf), F32)
    for (b, v), (off, w, ok) in tables.items():
        for k in range(4):
            contrib = (ds[b, v] * w[k][None]).astype(F32)      # zero where the tap is out of range / z <= 0
            for c in range(C):
                np.add.at(gf[b, v, c], off[k], contrib[c])
    return gf.reshape(features.shape)
