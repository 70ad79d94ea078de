"""Round 4: LDS bank-conflict model of the forward brick kernel's tap reads with PARITY-SPLIT windows.

A bilinear footprint {x0, x0+1} x {y0, y0+1} has exactly one pixel of each parity class (x & 1, y & 1).  If a view's window is
stored as four sub-images (one per class, each column-major over (x >> 1, y >> 1)), read instruction t of a view fetches class t
for every lane: lanes that are z-neighbours then hit the SAME or ADJACENT half-rows instead of rows up to 2 apart, and the 16
lanes of a ds_read_b128 group span half as many distinct slots.  Cycles per ds_read_b128 (4 = conflict-free), north-star geometry.
"""
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
grp = np.zeros(64, int); pos = np.zeros(64, int)
for gi, gl in enumerate(GROUPS):
    for k, ln in enumerate(gl): grp[ln] = gi; pos[ln] = k
# lane maps: (column of the wave's two, z inside the brick's 32)
MAPS = {
    "shipped (stride-4 transpose)": (h, 16 * g + 4 * b + a),
    "lane = z (no transpose)": (l >> 5, l & 31),
    "group = 16 consecutive z": (grp >> 1, (grp & 1) * 16 + pos),
    "group = 8 z col0 + same 8 z col1": (pos >> 3, grp * 8 + (pos & 7)),
}
BX, BY, BZ = 8, 8, 32
def run(sub=5, strides=("odd",)):
    acc = {}
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
                xm, ym = x0.min() & ~1, y0.min() & ~1
                bh = y0.max() + 1 - ym + 1
                hh = (bh + 1) // 2                                   # half-rows of a class image
                for u in range(2):
                  for w in range(16):
                    for nm, (cm, zm) in MAPS.items():
                        col = w * 2 + cm
                        cx, cy = (col & 3) + 4 * u, col >> 2
                        X = x0[cx, cy, zm] - xm; Y = y0[cx, cy, zm] - ym
                        # plain column-major (shipped layout)
                        tot = 0
                        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                            tot += cyc((X + dx) * (bh | 1) + Y + dy)
                        acc.setdefault((nm, "plain"), []).append(tot / 4)
                        for sname, st in (("hh", hh), ("hh|1", hh | 1), ("16k", (hh + 15) & ~15), ("8k", (hh + 7) & ~7), ("4k+2", ((hh + 3) & ~3) + 2)):
                            tot = 0
                            for px in (0, 1):
                              for py in (0, 1):
                                xt = X + ((X & 1) ^ px); yt = Y + ((Y & 1) ^ py)   # the tap of this parity class
                                tot += cyc((xt >> 1) * st + (yt >> 1))
                            acc.setdefault((nm, "parity " + sname), []).append(tot / 4)
    for k in sorted(acc):
        print("%-36s %-14s %.2f cycles per b128" % (k[0], k[1], np.mean(acc[k])))
if __name__ == "__main__":
    run()
