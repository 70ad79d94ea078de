"""How many 128-B lines (8 pixels of a column of the column-major quad-planar copy) does a brick's window cost per view, as the bounding
box of all taps (shipped) against per-column row spans (a window cut to each image column's own [ymin, ymax])?  North-star geometry."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
H = W = 96; S = 64; V = 4
BX, BY, BZ = 8, 8, 32
tot = {"bbox_slots": 0, "span_slots": 0, "bbox_lines": 0, "span_lines": 0, "touched_lines": 0}
for seed in (0,):
    P = bench.ring_projections(4, V, (H, W), seed=seed)
    coords = bench.cuboid_volume(1, S)[0]
    for b in range(4):
      for kx in range(0, S, BX):
        for ky in range(0, S, BY):
          for kz in range(0, S, BZ):
            pts = coords[kx:kx+BX, ky:ky+BY, kz:kz+BZ].reshape(-1, 3)
            hom = np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1)
            for v in range(V):
                r = hom @ P[b, v].T
                ix = (r[:, 0] / r[:, 2]) / H * (W - 1); iy = (r[:, 1] / r[:, 2]) / W * (H - 1)
                x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
                ok = (ix > -1) & (ix < W) & (iy > -1) & (iy < H)
                x0, y0 = x0[ok], y0[ok]
                if len(x0) == 0: continue
                xs = np.concatenate([x0, x0 + 1]); ys0 = np.concatenate([y0, y0]); ys1 = ys0 + 1
                xmin, xmax = xs.min(), xs.max(); ymin, ymax = ys0.min(), ys1.max()
                cols = xmax - xmin + 1
                # clamp to the image for the line count (rows outside are clamped duplicates)
                lo, hi = max(ymin, 0), min(ymax, H - 1)
                tot["bbox_slots"] += cols * (ymax - ymin + 1)
                tot["bbox_lines"] += cols * (hi // 8 - lo // 8 + 1)
                touched = set()
                for x in range(xmin, xmax + 1):
                    m = xs == x
                    a, bb = ys0[m].min(), ys1[m].max()
                    tot["span_slots"] += bb - a + 1
                    a2, b2 = max(a, 0), min(bb, H - 1)
                    tot["span_lines"] += b2 // 8 - a2 // 8 + 1
print(tot)
print("slots: per-column spans / bounding box = %.3f" % (tot["span_slots"] / tot["bbox_slots"]))
print("lines: per-column spans / bounding box = %.3f" % (tot["span_lines"] / tot["bbox_lines"]))
