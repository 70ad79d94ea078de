// Brick forward for 8 views at full occupancy: 1024 threads, one voxel per lane, 4 x 8 x 32 bricks, the views staged in GROUPS.
//
// k_fwd_brick serves 8 views with 512-thread blocks (a lane's 8 tap records, 2 x 32 sample registers and two tap sets need ~170
// VGPRs: two waves per SIMD) and 4 x 4 x 32 bricks (eight windows of a bigger brick overflow the LDS ring).  Here a ring item is
// (channel quad, view group): only the windows of VG = 4 views are resident at a time, so a 4 x 8 x 32 brick fits a 2-deep ring
// (group windows: mean 2 500, max 4 000 slots at the configs[3] geometry; 29 % fewer window pixels per voxel than 4 x 4 x 32), and a
// lane keeps one tap set and the 32 samples of its voxel: 128 VGPRs, four waves per SIMD.  The samples of both groups meet in
// registers and go through the same aggregate2<METHOD, 8> as in k_fwd_brick: identical results.
// Lane map, parity-split column-major windows and stores as k_fwd_brick (brick_fwd_kernel.h): fp32 volumes use the z-run map with four
// dword stores per job, 16-bit volumes the z-run map with the pair exchange.
#pragma once
#include "brick_fwd_kernel.h"

namespace mvhmr {

constexpr int kGroupViews = 4;
constexpr int kGroupChunks = 4;                       // DMA chunks a wave may own per ring item: 16 waves x 4 x 64 = 4 096 slots

// one voxel sampled straight from global memory: taps rebuilt from the projection (clamped taps, zero weights outside)
template <int METHOD, int VT, typename TO>
__device__ __attribute__((noinline)) void fwd_groups_slow(const float4 *fk, TO *obase, const float (*proj)[12], const Coords &coords, int b,
                                                          long long N, unsigned vox, int nq, int nqv, int H, int W, int nv)
{
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;
    const int HW = H * W;
    float c0, c1, c2;
    voxel_xyz(coords, b, N, vox, c0, c1, c2);
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const Taps t = make_taps(proj[v], c0, c1, c2, H, W);
        w00[v] = t.w00; w01[v] = t.w01; w10[v] = t.w10; w11[v] = t.w11;
        const int base = ((v < nv ? v : 0) * nqv) * HW;                         // an absent view reads view 0's pixels (and discards them)
        o00[v] = base + t.x0 * H + t.y0; o01[v] = base + t.x1 * H + t.y0; o10[v] = base + t.x0 * H + t.y1; o11[v] = base + t.x1 * H + t.y1;
    }
    for (int q = 0; q < nq; ++q) {
        const float4 *src = fk + (long long)q * HW;
        float s[4][VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const float4 a = src[o00[v]], bb = src[o01[v]], c = src[o10[v]], d = src[o11[v]];
            s[0][v] = bilerp(a.x, bb.x, c.x, d.x, w00[v], w01[v], w10[v], w11[v]);
            s[1][v] = bilerp(a.y, bb.y, c.y, d.y, w00[v], w01[v], w10[v], w11[v]);
            s[2][v] = bilerp(a.z, bb.z, c.z, d.z, w00[v], w01[v], w10[v], w11[v]);
            s[3][v] = bilerp(a.w, bb.w, c.w, d.w, w00[v], w01[v], w10[v], w11[v]);
            if (v >= nv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = kAbsentReads ? kAbsentSample : 0.f;
            }
        }
        TO *oq = obase + (long long)(q * 4) * N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r;
            if constexpr (METHOD == AGG_MEAN) r = __fdiv_rn(aggregate<AGG_SUM, VT>(s[i]), (float)nv);
            else r = aggregate<METHOD, VT>(s[i]);
            (oq + i * N)[vox] = from_f32<TO>(r);
        }
    }
}

template <int METHOD, int VT, typename TO>
__global__ void __launch_bounds__(1024)
k_fwd_brick_groups(const float4 *__restrict__ featK, const float *__restrict__ proj, const Coords coords, TO *__restrict__ out, int C,
                   int H, int W, int X, int Y, int Z, int nby, int nbz, int bricks_per_sample, int lds_slots, int total_blocks, int nv, int ksplit, Gate gate)
{
    // nv <= VT real views (5 ... 7 views run this kernel): the others are absent, as in k_fwd_brick
    if (gated_off(gate)) return;
    constexpr int NT = 1024, BY = NT / 128, NW = NT / 64, VG = kGroupViews, NG = VT / VG, MC = kGroupChunks;
    constexpr int LAY = kFwdLay, MAP = sizeof(TO) == 4 ? kFwdMapF32 : MVHMR_FWD_MAP16;
    static_assert(VT == 2 * VG, "two view groups");
    extern __shared__ __align__(16) unsigned char smem[];
    FwdShared<VT> *sh = reinterpret_cast<FwdShared<VT> *>(smem + lds_slots * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order as in k_fwd_brick
    const int nbx = bricks_per_sample / (nby * nbz);
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int tw = (nbx + tiles_x - 1) / tiles_x, th = (nby + tiles_y - 1) / tiles_y;
    const int share = tw * th * nbz;
    const int grid0 = (int)gridDim.x / ksplit, part = (int)blockIdx.x / grid0, bid = (int)blockIdx.x - part * grid0;   // channel split: k_fwd_brick
    const int xcd = bid & 7, j = bid >> 3;
    const int b = j / share, r = j % share;
    const int kz = r % nbz, cy = (r / nbz) % th, cx = r / (nbz * th);
    const int kx = (xcd % tiles_x) * tw + cx, ky = (xcd / tiles_x) * th + cy;
    if (kx >= nbx || ky >= nby || b * bricks_per_sample >= total_blocks) return;
    const long long N = (long long)X * Y * Z;
    const int HW = H * W, nqv = (C + 3) >> 2, nq = (C >> 2) / ksplit, q0 = part * nq;   // nqv: the copy's quads per view; nq: this block's WHOLE quads

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = tid < nv * 12 ? proj[((long long)b * nv) * 12 + tid] : 0.f;
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    __syncthreads();

    // ---- this lane's voxel and its tap records (once per brick)
    int dcol, zin;
    fwd_lane_voxel<MAP>(lane, dcol, zin);
    const int col = wave * 2 + dcol;
    const int vx_r = kx * kBX + (col & 3), vy_r = ky * BY + (col >> 2), vz_r = kz * kBZ + zin;
    const bool inside = vx_r < X && vy_r < Y && vz_r < Z;                           // volumes need not divide into bricks (brick_fwd_kernel.h)
    const int vx = vx_r < X ? vx_r : X - 1, vy = vy_r < Y ? vy_r : Y - 1, vz = vz_r < Z ? vz_r : Z - 1;
    const unsigned vox = (unsigned)(((long long)vx * Y + vy) * Z + vz);             // N < 2^28 (brick_fwd_supported)
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int tx[VT], ty[VT];
    unsigned valid = 0;
    {
        float c0, c1, c2;
        voxel_xyz(coords, b, N, vox, c0, c1, c2);
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
            w00[v] = t.w00; w01[v] = t.w01; w10[v] = t.w10; w11[v] = t.w11;
            tx[v] = t.rx0; ty[v] = t.ry0;
            const bool live = t.any && inside && v < nv;
            if (live) valid |= 1u << v;
            const int big = 1 << 30;
            const int nxmin = wave_max_dpp(live ? -t.rx0 : -big), nymin = wave_max_dpp(live ? -t.ry0 : -big);
            const int xmax = wave_max_dpp(live ? t.rx0 : -big), ymax = wave_max_dpp(live ? t.ry0 : -big);
            if (lane == 0 && xmax >= -nxmin) {
                atomicMin(&sh->bbox[v][0], -nxmin); atomicMin(&sh->bbox[v][1], -nymin);
                atomicMax(&sh->bbox[v][2], xmax); atomicMax(&sh->bbox[v][3], ymax);
            }
        }
    }
    __syncthreads();

    // ---- window per view (block-uniform); the views of a group are packed back to back, every group starts at slot 0
    // LAY 1: the rows of a window column split by parity (brick_fwd_kernel.h): origin row even, hp half-rows, column stride 2 hp
    int wx0[VT], wy0[VT], ws[VT], whp[VT], slot0[VT], nch[NG][VG + 1];
    int used = 0, max_stride = 0, max_chunks = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        int ug = 0;
        nch[g][0] = 0;
#pragma unroll
        for (int u = 0; u < VG; ++u) {
            const int v = g * VG + u;
            const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
            const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
            int bw = 0, stride = 1, hp = 0, y0w = ymin;
            if constexpr (LAY == 0) {
                int bh = 0;
                if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }
                stride = bh | 1;
            } else {
                y0w = ymin & ~1;
                if (xmax >= xmin) { bw = xmax - xmin + 2; hp = (ymax + 3 - y0w) >> 1; }
                stride = 2 * hp;
            }
            const int chunks = (stride * bw + 63) >> 6;
            wx0[v] = xmin; wy0[v] = y0w; ws[v] = stride; whp[v] = hp;
            max_stride = stride > max_stride ? stride : max_stride;
            slot0[v] = ug;
            ug += chunks << 6;
            nch[g][u + 1] = nch[g][u] + chunks;
        }
        used = ug > used ? ug : used;
        max_chunks = nch[g][VG] > max_chunks ? nch[g][VG] : max_chunks;
    }
    const int cap = fwd_cap2(lds_slots);
    const int buf_bytes = kZeroBytes + cap * 16;
    const bool fits = used <= cap && max_chunks <= MC * NW && max_stride + 2 <= kZeroSlots;
    TO *const obase = out + (long long)b * C * N + (long long)(q0 * 4) * N;
    const float4 *const fk = featK + (long long)b * nv * nqv * HW + (long long)q0 * HW;
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;
    constexpr bool kRel = METHOD == AGG_SOFTMAX;                                  // views >= 1 are folded as differences to view 0 (brick_fwd_kernel.h: ws_softmax_pair)

    if (fits) {
        for (int i = tid; i < kZeroSlots * 2; i += NT) {
            const float z = (kAbsentReads && nv < VT && i % kZeroSlots == kAbsentSlot) ? kAbsentSample : 0.f;
            *reinterpret_cast<float4 *>(smem + (i / kZeroSlots) * buf_bytes + (i % kZeroSlots) * 16) = make_float4(z, z, z, z);
        }
        int a0[VT], a1[LAY ? VT : 1], ws16[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            ws16[v] = ws[v] * 16;
            const bool ok = (valid >> v) & 1u;
            if constexpr (LAY == 0) {
                const int s0 = slot0[v] + (tx[v] - wx0[v]) * ws[v] + (ty[v] - wy0[v]);
                a0[v] = ok ? kZeroBytes + s0 * 16 : 0;
            } else {                                                             // even row, odd row; weights in that order (brick_fwd_kernel.h)
                const int yr = ty[v] - wy0[v];
                const int sc = slot0[v] + (tx[v] - wx0[v]) * ws[v];
                a0[v] = ok ? kZeroBytes + (sc + ((yr + 1) >> 1)) * 16 : 0;
                a1[v] = ok ? kZeroBytes + (sc + whp[v] + (yr >> 1)) * 16 : 0;
                if (yr & 1) {
                    const float t0 = w00[v], t1 = w01[v];
                    w00[v] = w10[v]; w01[v] = w11[v]; w10[v] = t0; w11[v] = t1;
                }
            }
            if (kAbsentReads && v >= nv) {                                       // the absent view's one "tap": kAbsentSample, weight 1
                a0[v] = kAbsentSlot * 16;
                if constexpr (LAY != 0) a1[v] = kAbsentSlot * 16;
                ws16[v] = 16;
                w00[v] = kRel ? 0.5f : 1.f; w01[v] = 0.f; w10[v] = 0.f; w11[v] = 0.f;   // (relative softmax: -FLT_MAX / 2 - s0 stays finite)
            }
        }
        // ---- DMA chunks of this wave, per group: chunk c covers 64 consecutive slots of one view's window
        unsigned g_off[NG][MC];
        int l_dst[NG][MC];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int rr = 0; rr < MC; ++rr) {
                const int c = wave + rr * NW;
                l_dst[g][rr] = -1;
                g_off[g][rr] = 0;
                if (c < nch[g][VG]) {
                    int u = 0;
#pragma unroll
                    for (int uu = 1; uu < VG; ++uu) u += c >= nch[g][uu] ? 1 : 0;
                    int sv = ws[g * VG], ox = wx0[g * VG], oy = wy0[g * VG], c0 = nch[g][0], s0 = slot0[g * VG], hv = whp[g * VG];
#pragma unroll
                    for (int uu = 1; uu < VG; ++uu)
                        if (u == uu) { sv = ws[g * VG + uu]; ox = wx0[g * VG + uu]; oy = wy0[g * VG + uu]; c0 = nch[g][uu]; s0 = slot0[g * VG + uu]; hv = whp[g * VG + uu]; }
                    const int jj = c - c0, slot = (jj << 6) + lane;
                    const int px = slot / sv;
                    int py = slot - px * sv;
                    if constexpr (LAY == 1) py = py >= hv ? 2 * (py - hv) + 1 : 2 * py;
                    int gx = ox + px, gy = oy + py;
                    gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
                    gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
                    g_off[g][rr] = (unsigned)(((g * VG + u) * nqv) * HW + gx * H + gy) * 16u;
                    l_dst[g][rr] = kZeroBytes + (s0 + (jj << 6)) * 16;
                }
            }
        const unsigned lds_base = (unsigned)(size_t)(lds_void_t *)smem;
        // ring item t = NG * quad + group, buffer t & 1
        auto dma = [&](int q, auto gtag) __attribute__((always_inline)) {
            constexpr int g = decltype(gtag)::value;
            const float4 *src = fk + (long long)q * HW;
            const int boff = ((q * NG + g) & 1) * buf_bytes;
#pragma unroll
            for (int rr = 0; rr < MC; ++rr)
                if (l_dst[g][rr] >= 0) glds16(src, g_off[g][rr], lds_base + (unsigned)uniform(l_dst[g][rr] + boff));
        };

        constexpr unsigned OSZ = sizeof(TO);
        const unsigned chan_bytes = (unsigned)(N * OSZ);
        const int z0 = ((lane >> 5) << 4) + ((lane & 3) << 2);
        static_assert(MAP == 1, "the stride-4 transpose map writes four z per lane: whole bricks only");
        const unsigned st_off = !inside ? kDropOffset                                // beyond num_records: the stores are dropped
                              : MAP == 1 ? (OSZ == 4 ? vox * OSZ : (vox - (unsigned)(lane & 1)) * OSZ + (unsigned)(lane & 1) * 2u * chan_bytes)
                                         : (vox - (unsigned)zin + (unsigned)z0) * OSZ + (unsigned)((lane >> 2) & 3) * chan_bytes;
        auto store_quad = [&](int q, float (&res)[4]) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + (long long)(q * 4) * N, 0, (int)(4u * chan_bytes), 0x00020000);
            if constexpr (MAP == 1 && OSZ == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[i]), rs, (int)st_off, (int)(i * chan_bytes), kStAux);
                return;
            } else if constexpr (MAP == 1) {                                     // 16-bit volume: pair exchange (brick_fwd_kernel.h)
                const bool odd = lane & 1;
                const float s0 = odd ? res[0] : res[2], s1 = odd ? res[1] : res[3];
                const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
                const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
                const unsigned d0 = odd ? pack2<TO>(g0, res[2]) : pack2<TO>(res[0], g0);
                const unsigned d1 = odd ? pack2<TO>(g1, res[3]) : pack2<TO>(res[1], g1);
                __builtin_amdgcn_raw_buffer_store_b32(d0, rs, (int)st_off, 0, kStAux);
                __builtin_amdgcn_raw_buffer_store_b32(d1, rs, (int)st_off, (int)chan_bytes, kStAux);
                return;
            }
            stride4_transpose(res, lane);
            if constexpr (OSZ == 4) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 d = {__builtin_bit_cast(unsigned, res[0]), __builtin_bit_cast(unsigned, res[1]),
                                 __builtin_bit_cast(unsigned, res[2]), __builtin_bit_cast(unsigned, res[3])};
                __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)st_off, 0, kStAux);
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 d = {pack2<TO>(res[0], res[1]), pack2<TO>(res[2], res[3])};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, (int)st_off, 0, kStAux);
            }
        };

        float s[4][VT];
        dma(0, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the zero regions are written
        // item (q, g): wait for its DMA (issued one item earlier; at most the store of the previous quad is younger), barrier
        // (publishes the item; every wave has folded the item before, whose buffer is therefore free), request the next item,
        // fold the group's views; after the last group aggregate, transpose, store
#pragma nounroll
        for (int q = 0; q < nq; ++q) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                constexpr int SPJ = MAP == 1 ? (OSZ == 4 ? 4 : 2) : 1;           // store instructions per quad
                if (g == 0 && q > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SPJ) : "memory");   // g == 0: the stores of quad q-1 were issued after this DMA
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bare_barrier();
                if (g + 1 < NG) {
                    if constexpr (NG > 1) dma(q, std::integral_constant<int, (NG > 1 ? 1 : 0)>{});
                } else if (q + 1 < nq) {
                    dma(q + 1, std::integral_constant<int, 0>{});
                }
                const int boff = ((q * NG + g) & 1) * buf_bytes;
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    const int v = g * VG + u;
                    int base = a0[v] + boff;
                    asm volatile("" : "+v"(base));                               // rebuilt per use: hoisted, the 16 addresses would spill
                    const int far = base + ws16[v];
                    f32x4 nw, sw, ne, se;                                        // LAY 1: even row x0, odd row x0, even row x0+1, odd row x0+1
                    if constexpr (LAY == 0) {
                        nw = lds_tap(smem, base); sw = lds_tap(smem, base + 16); ne = lds_tap(smem, far); se = lds_tap(smem, far + 16);
                    } else {
                        int base1 = a1[v] + boff;
                        asm volatile("" : "+v"(base1));
                        nw = lds_tap(smem, base); sw = lds_tap(smem, base1); ne = lds_tap(smem, far); se = lds_tap(smem, base1 + ws16[v]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (kRel && v > 0) s[i][v] = bilerp_rel(nw.v[i], ne.v[i], sw.v[i], se.v[i], w00[v], w01[v], w10[v], w11[v], s[i][0]);
                        else s[i][v] = bilerp(nw.v[i], ne.v[i], sw.v[i], se.v[i], w00[v], w01[v], w10[v], w11[v]);
                        asm volatile("" : "+v"(s[i][v]));
                    }
                }
            }
            float res[4];
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                if constexpr (kRel) {
                    // softmax relative to view 0 (ws_softmax_pair: the views >= 1 were folded as differences), overflow test on the denominators,
                    // the max form as the wave-uniform fallback -- no maximum over the eight views, no subtractions, one exponential fewer
                    float d;
                    ws_softmax_pair<VT, false>(s[i], s[i + 1], res[i], res[i + 1], d);
                    if (__builtin_amdgcn_ballot_w64(!(d < 1.152921504606847e18f)) != 0) {
                        res[i] = ws_softmax_safe<VT, false>(s[i]);
                        res[i + 1] = ws_softmax_safe<VT, false>(s[i + 1]);
                    }
                } else {
                    fwd_aggregate2<METHOD, VT>(s[i], s[i + 1], res[i], res[i + 1], (float)nv);
                }
            }
            store_quad(q, res);
        }
    } else {
        // ---- windows do not fit the LDS pool: sample straight from global memory (its own function: its registers -- 8 views of
        // samples, weights and 64-bit addresses -- stay out of the fast path's allocation)
        if (inside) fwd_groups_slow<METHOD, VT, TO>(fk, obase, sh->proj, coords, b, N, vox, nq, nqv, H, W, nv);
    }
}

template <int METHOD, int VT, typename TO>
hipError_t launch_fwd_groups_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s)
{
    constexpr int NT = 1024;
    const int nbx = (p.X + kBX - 1) / kBX, nby = (p.Y + NT / 128 - 1) / (NT / 128), nbz = (p.Z + kBZ - 1) / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int slots = fwd_lds_slots();
    const size_t lds = (size_t)slots * 16 + sizeof(FwdShared<VT>);
    auto kern = k_fwd_brick_groups<METHOD, VT, TO>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int grid = ((nbx + tiles_x - 1) / tiles_x) * ((nby + tiles_y - 1) / tiles_y) * nbz * 8 * p.B;
    const int ks = brick_fwd_ksplit(total, p.C / 4);
    hipLaunchKernelGGL(kern, dim3(grid * ks), dim3(NT), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, slots, total, p.V,
                       ks, make_gate(p, true));
    return hipGetLastError();
}

}  // namespace mvhmr
