import sys, numpy as np
sys.path.insert(0, '/root/repo')
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
def cyc(slots):
    c = 0
    for g in GROUPS:
        a = slots[g]; cls = a % 16
        worst = 1
        for s in np.unique(cls):
            worst = max(worst, len(np.unique(a[cls == s])))
        c += worst
    return c
l = np.arange(64)
b, a, h, g = l & 3, (l >> 2) & 3, (l >> 4) & 1, l >> 5
zmap = 16 * g + 4 * b + a; cmap = h          # shipped: column h of the wave's two, z = 16g + 4b + a
BX, BY, BZ = 8, 8, 32
def run(layout_fn, names, sub=5):
    acc = {n: [] for n in names}
    n = 0
    for kx in range(0, S, BX):
      for ky in range(0, S, BY):
        for kz in range(0, S, BZ):
            n += 1
            if n % sub: continue
            pts = coords[kx:kx+BX, ky:ky+BY, kz:kz+BZ]
            for v in range(V):
                x0, y0 = taps(P[v], pts.reshape(-1, 3))
                x0 = x0.reshape(BX, BY, BZ); y0 = y0.reshape(BX, BY, BZ)
                xm, ym = x0.min(), y0.min()
                bw, bh = x0.max() - xm + 2, y0.max() - ym + 2
                for u in range(2):
                  for w in range(16):
                    col = w * 2 + cmap
                    cx, cy = (col & 3) + 4 * u, col >> 2
                    X = x0[cx, cy, zmap] - xm; Y = y0[cx, cy, zmap] - ym
                    for nm in names:
                        tot = 0
                        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                            tot += cyc(layout_fn(nm, X + dx, Y + dy, bw, bh))
                        acc[nm].append(tot / 4)
    for nm in names:
        print("%-28s %.2f cycles per b128" % (nm, np.mean(acc[nm])))
def lay(nm, xx, yy, bw, bh):
    kind, p = nm
    if kind == "rot16": return xx * (bh | 1) + yy + p * (yy >> 4)
    if kind == "rot8": return xx * (bh | 1) + yy + p * (yy >> 3)
    if kind == "stride": return xx * (((bh + p) | 1)) + yy
    if kind == "xrot": return xx * (bh | 1) + yy + p * (xx & 1)
names = [("rot16", g_) for g_ in range(0, 16)] + [("rot8", g_) for g_ in (1, 2, 3, 5, 7)] + [("xrot", p) for p in (4, 8)]
run(lay, names)

print("---- lane maps (col-major, stride bh|1)")
def run_maps(maps, sub=5):
    acc = {n: [] for n in maps}
    n = 0
    for kx in range(0, S, BX):
      for ky in range(0, S, BY):
        for kz in range(0, S, BZ):
            n += 1
            if n % sub: continue
            pts = coords[kx:kx+BX, ky:ky+BY, kz:kz+BZ]
            for v in range(V):
                x0, y0 = taps(P[v], pts.reshape(-1, 3))
                x0 = x0.reshape(BX, BY, BZ); y0 = y0.reshape(BX, BY, BZ)
                xm, ym = x0.min(), y0.min()
                bh = y0.max() - ym + 2
                for u in range(2):
                  for w in range(16):
                    for nm, (cm, zm) in maps.items():
                        col = w * 2 + cm
                        cx, cy = (col & 3) + 4 * u, col >> 2
                        X = x0[cx, cy, zm] - xm; Y = y0[cx, cy, zm] - ym
                        tot = 0
                        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                            tot += cyc((X + dx) * (bh | 1) + Y + dy)
                        acc[nm].append(tot / 4)
    for nm in maps:
        print("%-44s %.2f cycles per b128" % (nm, np.mean(acc[nm])))
maps = {"shipped": (cmap, zmap)}
# position of each lane inside its group / which group
grp = np.zeros(64, int); pos = np.zeros(64, int)
for gi, gl in enumerate(GROUPS):
    for k, ln in enumerate(gl): grp[ln] = gi; pos[ln] = k
maps["group = 16 consecutive z of one column"] = (grp >> 1, (grp & 1) * 16 + pos)
maps["group = 8 z of col0 + same 8 z of col1"] = (pos >> 3, grp * 8 + (pos & 7))
maps["group = 8 z col0 + 8 z col1, interleaved"] = (pos & 1, grp * 8 + (pos >> 1))
maps["group = even z of col0 + even z col1 (stride 2)"] = (pos >> 3, (grp & 1) + 2 * (pos & 7) + 16 * (grp >> 1))
run_maps(maps)
