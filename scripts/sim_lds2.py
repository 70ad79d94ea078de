"""Search lane->voxel mappings / LDS row strides for the brick kernel's ds_read_b128 conflict factor (offline model)."""
import sys, numpy as np, itertools
sys.path.insert(0, '.')
import bench
H = W = 96; S = 64; V = 4
P = bench.ring_projections(1, V, (H, W), seed=0)[0]
coords = bench.cuboid_volume(1, S)[0]
def taps(Pv, pts):
    hom = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1)
    r = hom @ Pv.T
    ix = (r[:, 0] / r[:, 2]) / H * (W - 1); iy = (r[:, 1] / r[:, 2]) / W * (H - 1)
    ok = (ix > -1) & (ix < W) & (iy > -1) & (iy < H)
    return np.floor(ix).astype(int), np.floor(iy).astype(int), ok
G0 = [0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]; G1 = [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
GROUPS = [G0, G1, [l+32 for l in G0], [l+32 for l in G1]]
def cyc(slots):
    c = 0
    for g in GROUPS:
        a = slots[g]; worst = 1
        for s in range(16):
            worst = max(worst, len(set(a[a % 16 == s])))
        c += worst
    return c
BX, BY, BZ = 4, 8, 32
def evaluate(lanemap, stride_fn, label, sub=11):
    """lanemap: list of 64 (dcol, z) : lane -> (column offset within the wave's 2 columns, z in 0..31)"""
    tot = []; n = 0
    for kx in range(0, S, BX):
        for ky in range(0, S, BY):
            for kz in range(0, S, BZ):
                n += 1
                if n % sub: continue
                pts = coords[kx:kx+BX, ky:ky+BY, kz:kz+BZ].reshape(-1, 3)
                for v in range(V):
                    x0, y0, ok = taps(P[v], pts)
                    if not ok.any(): continue
                    xmin, ymin = x0[ok].min(), y0[ok].min()
                    bw = x0[ok].max() - xmin + 2
                    st = stride_fn(bw)
                    px, py = x0 - xmin, y0 - ymin
                    for w in range(0, BX * BY, 2):
                        lanes = np.empty(64, int)
                        for l, (dc, z) in enumerate(lanemap):
                            col = w + dc; cx, cy = col % BX, col // BX
                            idx = (cx * BY + cy) * BZ + z
                            lanes[l] = py[idx] * st + px[idx]
                        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                            tot.append(cyc(lanes + dy * st + dx))
    print("%-52s %.2f cycles per b128 (ideal 4)" % (label, np.mean(tot)))
    return np.mean(tot)
odd = lambda bw: bw | 1
def map_ident():  return [(l >> 5, l & 31) for l in range(64)]
def map_groupcontig():
    m = [None]*64
    for half in range(2):
        for i, l in enumerate(G0): m[half*32 + l] = (half, i)
        for i, l in enumerate(G1): m[half*32 + l] = (half, 16 + i)
    return m
def map_8x2():
    # each b128 group = 8 consecutive z of column 0 and the same 8 z of column 1
    m = [None]*64
    for gi, g in enumerate(GROUPS):
        for i, l in enumerate(g): m[l] = (i >> 3, gi * 8 + (i & 7))
    return m
def map_4x4_strided():
    # each group: 16 consecutive z but interleaved so that ... (control)
    return map_groupcontig()
evaluate(map_ident(), odd, "identity, stride bw|1")
evaluate(map_groupcontig(), odd, "group = 16 consecutive z, stride bw|1")
evaluate(map_8x2(), odd, "group = 8 z x 2 columns, stride bw|1")
for k in (1, 3, 5, 7, 9, 11, 13, 15):
    evaluate(map_8x2(), (lambda k: lambda bw: bw + ((k - bw) % 16))(k), "group = 8 z x 2 columns, stride == %d mod 16" % k)
for k in (3, 5, 7, 9, 11, 13):
    evaluate(map_groupcontig(), (lambda k: lambda bw: bw + ((k - bw) % 16))(k), "group = 16 z, stride == %d mod 16" % k)
