"""Y-only parity split, exact layout of the r04 kernel: slot(X, y) = X * S + (y & 1) * hp + (y >> 1), S = 2 * hp,
hp = roundup(ceil(rows / 2), R); window origin y even.  Instructions: (x0, even row), (x0, odd row), (x0+1, even), (x0+1, odd)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from sim_lds5 import taps, cyc, GROUPS, MAPS, P, coords, S, V, BX, BY, BZ
def run(sub=3):
    acc = {}; slots = {}
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
                xm, ym = x0.min(), y0.min() & ~1
                bw = x0.max() - xm + 2
                rows = y0.max() + 1 - ym + 1
                hh = (rows + 1) // 2
                slots.setdefault("plain", []).append(bw * ((rows) | 1))
                for R in (1, 2, 4, 8):
                    hp = (hh + R - 1) // R * R
                    slots.setdefault("R%d" % R, []).append(bw * 2 * hp)
                for u in range(2):
                  for w in range(16):
                    for nm, (cm, zm) in MAPS.items():
                        col = w * 2 + cm
                        cx, cy = (col & 3) + 4 * u, col >> 2
                        X = x0[cx, cy, zm] - xm; Y = y0[cx, cy, zm] - ym
                        for R in (1, 2, 4, 8):
                            hp = (hh + R - 1) // R * R
                            tot = 0
                            for dx in (0, 1):
                              for par in (0, 1):
                                yt = Y + ((Y & 1) ^ par)
                                tot += cyc((X + dx) * 2 * hp + par * hp + (yt >> 1))
                            acc.setdefault((nm, "R=%d" % R), []).append(tot / 4)
    for k in sorted(acc):
        print("%-36s %-8s %.2f cycles per b128" % (k[0], k[1], np.mean(acc[k])))
    for k in slots: print("window slots per view, mean %-6s %.0f  max %d" % (k, np.mean(slots[k]), np.max(slots[k])))
if __name__ == "__main__":
    run()
