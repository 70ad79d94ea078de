"""Offline model: ds_read_b128 bank-conflict cycles for lane->voxel maps x LDS window layouts (north-star geometry)."""
import sys, numpy as np
sys.path.insert(0, '.')
import bench
H = W = 96; S = 64; V = 4
P = bench.ring_projections(1, V, (H, W), seed=0)[0]
coords = bench.cuboid_volume(1, S)[0]
G0 = [0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]; G1 = [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
GROUPS = np.array([G0, G1, [l+32 for l in G0], [l+32 for l in G1]])
def taps(Pv, pts):
    hom = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1)
    r = hom @ Pv.T
    ix = (r[:, 0] / r[:, 2]) / H * (W - 1); iy = (r[:, 1] / r[:, 2]) / W * (H - 1)
    return np.floor(ix).astype(int), np.floor(iy).astype(int)
def cyc(slots):            # slots: (64,) int -> cycles
    c = 0
    for g in GROUPS:
        a = slots[g]; cls = a % 16
        worst = 1
        for s in np.unique(cls):
            worst = max(worst, len(np.unique(a[cls == s])))
        c += worst
    return c
def lane_maps():
    m = {}
    l = np.arange(64)
    # current kernel: col = l>>5, z via zin table
    l5 = l & 31
    zin = np.where(l5 < 4, l5, np.where(l5 < 12, 12 + l5, np.where(l5 < 16, l5 - 8, np.where(l5 < 20, 8 + l5, np.where(l5 < 28, l5 - 12, l5)))))
    m["current (16 consecutive z per group)"] = (l >> 5, zin)
    b, a, h, g = l & 3, (l >> 2) & 3, (l >> 4) & 1, l >> 5
    m["M1 z=4b+a+16h col=g"] = (g, 4 * b + a + 16 * h)
    m["M2 z=4b+a+16g col=h"] = (h, 4 * b + a + 16 * g)
    j, k = l & 15, l >> 4
    m["permlane swap: col=j>>3 z=4(j&7)+k"] = (j >> 3, 4 * (j & 7) + k)
    return m
def evaluate(BX, BY, BZ, sub=13):
    res = {}
    maps = lane_maps()
    layouts = {"row-major stride bw|1": 0, "col-major stride bh|1": 1, "row-major bw|1 + 5*(r>>4)": 2, "col-major bh|1 +3*(r>>4)": 3}
    acc = {(mn, ln): [] for mn in maps for ln in layouts}
    n = 0
    for kx in range(0, S, BX):
        for ky in range(0, S, BY):
            for kz in range(0, S, BZ):
                n += 1
                if n % sub: continue
                pts = coords[kx:kx+BX, ky:ky+BY, kz:kz+BZ].reshape(-1, 3)
                for v in range(V):
                    x0, y0 = taps(P[v], pts)
                    px, py = x0 - x0.min(), y0 - y0.min()
                    bw, bh = px.max() + 2, py.max() + 2
                    for w in range(0, BX * BY, 2):
                        for mn, (cm, zm) in maps.items():
                            col = w + cm; cx, cy = col % BX, col // BX
                            idx = (cx * BY + cy) * BZ + zm
                            X, Y = px[idx], py[idx]
                            for ln, lt in layouts.items():
                                tot = 0
                                for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                                    xx, yy = X + dx, Y + dy
                                    if lt == 0: sl = yy * (bw | 1) + xx
                                    elif lt == 1: sl = xx * (bh | 1) + yy
                                    elif lt == 2: sl = yy * (bw | 1) + xx + 5 * (yy >> 4)
                                    else: sl = xx * (bh | 1) + yy + 3 * (yy >> 4)
                                    tot += cyc(sl)
                                acc[(mn, ln)].append(tot / 4)
    for k, v in acc.items():
        print("%-40s | %-28s : %.2f cycles per b128" % (k[0], k[1], np.mean(v)))
evaluate(4, 8, 32)
