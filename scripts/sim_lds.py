"""Offline model of the brick kernel's LDS behaviour on the benchmark geometry:
   (1) distribution of the pooled window size per brick, (2) ds_read_b128 conflict factor per lane mapping."""
import sys, numpy as np
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
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS += [[l+32 for l in g] for g in GROUPS]
def conflict_cycles(slots_per_lane):      # 64 lanes -> LDS cycles for one ds_read_b128 (4 ideal)
    cyc = 0
    for g in GROUPS:
        a = slots_per_lane[g]
        worst = 1
        for s in range(16):
            worst = max(worst, len(set(a[a % 16 == s])))
        cyc += worst
    return cyc
def run(BX, BY, BZ, zmap, stride_fn, label):
    sizes = []; cyc = []; n = 0
    for kx in range(0, S, BX):
        for ky in range(0, S, BY):
            for kz in range(0, S, BZ):
                pts = coords[kx:kx+BX, ky:ky+BY, kz:kz+BZ].reshape(-1, 3)     # order: x, y, z
                tot = 0; info = []
                for v in range(V):
                    x0, y0, ok = taps(P[v], pts)
                    if not ok.any(): info.append(None); continue
                    xmin, xmax, ymin, ymax = x0[ok].min(), x0[ok].max(), y0[ok].min(), y0[ok].max()
                    bw, bh = xmax - xmin + 2, ymax - ymin + 2
                    st = stride_fn(bw)
                    tot += -(-(st * bh) // 64) * 64
                    info.append((x0 - xmin, y0 - ymin, st, ok))
                sizes.append(tot)
                if n % 7 == 0:        # conflict model on a subsample of bricks
                    ncol = BX * BY
                    for w in range(0, ncol, 2):                   # a wave = 2 columns x 32 z
                        for v in range(V):
                            if info[v] is None: continue
                            px, py, st, ok = info[v]
                            lanes = np.empty(64, int)
                            for half in range(2):
                                col = w + half
                                cx, cy = col % BX, col // BX                     # col&3 = x, col>>2 = y  (kernel mapping)
                                for l in range(32):
                                    z = zmap[l]
                                    idx = (cx * BY + cy) * BZ + z
                                    lanes[half * 32 + l] = py[idx] * st + px[idx]
                            for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                                cyc.append(conflict_cycles(lanes + dy * st + dx))
                n += 1
    sizes = np.array(sizes)
    print("%-34s window slots: mean %5.0f p50 %5.0f p90 %5.0f p99 %5.0f max %5.0f | frac<=3328 %.3f <=5056 %.3f | b128 cycles mean %.2f (ideal 4)" % (
        label, sizes.mean(), np.percentile(sizes, 50), np.percentile(sizes, 90), np.percentile(sizes, 99), sizes.max(),
        (sizes <= 3328).mean(), (sizes <= 5056).mean(), np.mean(cyc)))
ident = list(range(32))
# group-contiguous z: b128 lane groups {0-3,12-15,20-27} -> z 0..15 ; {4-11,16-19,28-31} -> z 16..31
gc = [0]*32
for i, l in enumerate([0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]): gc[l] = i
for i, l in enumerate([4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]): gc[l] = 16 + i
# interleaved: group A gets even z, group B odd z
il = [0]*32
for i, l in enumerate([0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]): il[l] = 2*i
for i, l in enumerate([4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]): il[l] = 2*i+1
odd = lambda bw: bw | 1
run(4, 8, 32, ident, odd, "4x8x32 identity, odd stride")
run(4, 8, 32, gc, odd, "4x8x32 group-contiguous z")
run(4, 8, 32, il, odd, "4x8x32 interleaved z")
run(4, 8, 32, gc, lambda bw: bw, "4x8x32 group-contig, raw stride")
for k in (3, 5, 7, 9):
    run(4, 8, 32, gc, (lambda k: lambda bw: ((bw + 15) // 16) * 16 + k)(k), "4x8x32 gc, stride = 16n+%d" % k)
run(4, 4, 32, gc, odd, "4x4x32 group-contiguous z")
