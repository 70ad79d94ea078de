// Fused un-projection forward, "brick" variant: LDS-staged feature windows (gfx950 / CDNA4 only).
//
// Reference semantics: models/aggregation.py:20-87 (projection, depth mask, bilinear grid_sample with zero padding,
// cross-view aggregate); the arithmetic order is pinned in device_common.h.
//
// Mapping.
//   block   = one brick of (4 * NVOX) x (NT / 128) x 32 voxels of one sample, NT threads; a lane owns NVOX voxels (x, x+4),
//             whose tap records (LDS address + 4 weights per view) are computed once and live in registers while the block
//             loops over the C / 4 channel quads;
//   windows = per view, the bounding box of the brick's taps; the views' windows are packed into one LDS buffer;
//             the staged copy of the features and the LDS image are COLUMN-major quad-planar (B,V,C/4,Wf,Hf,4): a z-long
//             brick seen by an upright camera gives tall narrow windows, so a window column is one contiguous run of the
//             staged copy -- 26 % fewer 128-B line fills than row-major (profiles/r02_fwd_ablations.txt);
//   ring    = 2 or 3 buffers filled by LDS-DMA (global_load_lds_dwordx4).  Top of quad q: wait for this wave's DMA of quad q,
//             s_barrier (publishes quad q; every wave has folded quad q-1, so that buffer is free), DMA of the next quad;
//   jobs    = (quad q, voxel u): request views 0 and 1, aggregate half of the previous job, fold view 0 / request view 2,
//             fold view 1 / request view 3, aggregate the other half + transpose + store, fold views 2 and 3 -- LDS reads,
//             FMAs, transcendentals and the store are spread over the job;
//   stores  = the 4 channels x 4 voxels of lanes {l, l^4, l^8, l^12} are transposed with DPP row shifts under bank masks, so
//             that a lane QUAD writes 64 contiguous bytes of one channel: one TCP access per 4 lanes (the r01 kernel's in-quad
//             transpose made every lane its own 16-B access: 64 per store instruction, and the CU's vector-memory pipe is
//             what bounds this kernel).
// Two voxels per lane (8 x 8 x 32 bricks) cut the window bytes per voxel by a third; the price is a 2-deep ring.
#pragma once
#include "brick_common.h"
#include "kernels.h"

namespace mvhmr {

// Timing-only ablations for scripts/exp (never defined in the product build): bit 0 conflict-free fake tap addresses, 1 no tap reads,
// 2 no LDS-DMA, 3 no stores, 4 no transcendentals, 5 no transpose, 6 no per-quad barrier, 7 no aggregate, 8 no wait for the DMA,
// 9 LDS-DMA without the m0 save / restore, 10 phase timers of a brick (s_memtime, summed over all waves: mvhmr_exp_fwd_timers_read)
#ifndef MVHMR_EXP
#define MVHMR_EXP 0
#endif
constexpr int kExp = MVHMR_EXP;
#if MVHMR_EXP & 1024
__device__ unsigned long long g_exp_fwd_timers[8];
#define EXP_FT(i) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define EXP_FT(i) do { } while (0)
#endif

// lane = 32 g + 16 h + 4 a + b  ->  column h of the wave's two (x-adjacent) columns, z = 16 g + 4 b + a
__device__ __forceinline__ void fwd_lane_voxel(int lane, int &dcol, int &zin)
{
    dcol = (lane >> 4) & 1;
    zin = ((lane >> 5) << 4) + ((lane & 3) << 2) + ((lane >> 2) & 3);
}

// DPP move under a bank mask (bank k = lanes 4k..4k+3 of every 16-lane row): masked-off lanes keep `keep`
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_into(float keep, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, x), CTRL, 0xF, BANKS, false));
}

// 4 x 4 transpose across lanes {l, l^4, l^8, l^12} of a 16-lane row: on entry lane 4a+b holds r[i] = value(channel i, z_a);
// on exit it holds r[k] = value(channel a, z_k).  Two butterfly stages of 2 selects + 4 masked DPP moves.
__device__ __forceinline__ void stride4_transpose(float (&r)[4], int lane)
{
    const bool a0 = lane & 4, a1 = lane & 8;
    constexpr int SHL4 = 0x104, SHR4 = 0x114, ROR8 = 0x128;                      // row_shl:4 (from lane+4), row_shr:4 (from lane-4), row_ror:8
    {
        const float x = a0 ? r[0] : r[1], y = a0 ? r[2] : r[3];                  // what the partner (lane ^ 4) takes
        const float n1 = dpp_into<SHL4, 0x5>(r[1], x), n0 = dpp_into<SHR4, 0xA>(r[0], x);
        const float n3 = dpp_into<SHL4, 0x5>(r[3], y), n2 = dpp_into<SHR4, 0xA>(r[2], y);
        r[0] = n0; r[1] = n1; r[2] = n2; r[3] = n3;
    }
    {
        const float x = a1 ? r[0] : r[2], y = a1 ? r[1] : r[3];                  // partner = lane ^ 8
        const float n2 = dpp_into<ROR8, 0x3>(r[2], x), n0 = dpp_into<ROR8, 0xC>(r[0], x);
        const float n3 = dpp_into<ROR8, 0x3>(r[3], y), n1 = dpp_into<ROR8, 0xC>(r[1], y);
        r[0] = n0; r[1] = n1; r[2] = n2; r[3] = n3;
    }
}

// aggregate<> behind the timing-only ablations (kExp == 0: exactly aggregate<>)
template <int METHOD, int VT>
__device__ __forceinline__ float fwd_aggregate(const float (&s)[VT])
{
    if constexpr (kExp & 128) {
        float r = s[0];
#pragma unroll
        for (int v = 1; v < VT; ++v) r += s[v];
        return r;
    } else if constexpr ((kExp & 16) && METHOD == AGG_SOFTMAX && VT == 4) {
        const float m = vmax(vmax3(s[0], s[1], s[2]), s[3]);
        const float nm = -m * 1.4426950408889634f;
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            float e = fmaf(s[v], 1.4426950408889634f, nm);
            e = fmaf(e, 0.5f, 1.f);                                              // stands in for v_exp_f32
            den += e;
            num = fmaf(e, s[v], num);
        }
        return num * fmaf(den, 0.25f, 1.f);                                      // stands in for v_rcp_f32
    } else {
        return aggregate<METHOD, VT>(s);
    }
}

// aggregate2<> behind the timing-only ablations
template <int METHOD, int VT>
__device__ __forceinline__ void fwd_aggregate2(const float (&sa)[VT], const float (&sb)[VT], float &ra, float &rb)
{
    if constexpr (kExp & (16 | 128)) {
        ra = fwd_aggregate<METHOD, VT>(sa);
        rb = fwd_aggregate<METHOD, VT>(sb);
    } else {
        aggregate2<METHOD, VT>(sa, sb, ra, rb);
    }
}

template <int VT>
struct FwdShared {
    int bbox[VT][4];               // xmin, ymin, xmax, ymax of the nw taps (valid voxels only)
    float proj[VT][12];
};

// pool geometry shared by the kernel, the gate and the host: `slots` 16-B slots hold nb buffers of kZeroSlots + cap slots
__host__ __device__ inline int fwd_cap3(int slots) { return ((slots - 3 * kZeroSlots) / 3) & ~63; }
__host__ __device__ inline int fwd_cap2(int slots) { return ((slots - 2 * kZeroSlots) / 2) & ~63; }

template <int METHOD, int VT, int NT, typename TO, int NVOX>
__global__ void __launch_bounds__(NT)
k_fwd_brick(const float4 *__restrict__ featK, const float *__restrict__ proj, const Coords coords,
            TO *__restrict__ out, int C, int H, int W, int X, int Y, int Z, int nby, int nbz, int bricks_per_sample,
            int lds_slots, int total_blocks, Gate gate)
{
    if (gated_off(gate)) return;
#if MVHMR_EXP & 1024
    unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif
    constexpr int BY = NT / 128, NW = NT / 64, BXK = kBX * NVOX;
    constexpr int MC = brick_chunks_per_wave(NT);                                 // DMA chunks a wave may own per quad
    extern __shared__ __align__(16) unsigned char smem[];
    FwdShared<VT> *sh = reinterpret_cast<FwdShared<VT> *>(smem + lds_slots * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order (speed only).  Blocks i, i+8, ... share an XCD under round-robin dispatch.  All eight XCDs work on the
    // same sample at a time, each on a compact tile of brick columns (all z): neighbouring windows meet in one L2 and the live
    // feature planes stay in the Infinity Cache.
    const int nbx = bricks_per_sample / (nby * nbz);
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int tw = (nbx + tiles_x - 1) / tiles_x, th = (nby + tiles_y - 1) / tiles_y;
    const int share = tw * th * nbz;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = j / share, r = j % share;
    const int kz = r % nbz, cy = (r / nbz) % th, cx = r / (nbz * th);
    const int kx = (xcd % tiles_x) * tw + cx, ky = (xcd / tiles_x) * th + cy;
    if (kx >= nbx || ky >= nby || b * bricks_per_sample >= total_blocks) return;
    const long long N = (long long)X * Y * Z;
    const int HW = H * W, nq = C >> 2;

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = proj[((long long)b * VT) * 12 + tid];
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    __syncthreads();

    // ---- this lane's voxels and their tap records (once per brick)
    int dcol, zin;
    fwd_lane_voxel(lane, dcol, zin);
    const int col = wave * 2 + dcol;
    const int vy = ky * BY + (col >> 2), vz = kz * kBZ + zin;
    unsigned vox[NVOX];
    float w00[NVOX][VT], w01[NVOX][VT], w10[NVOX][VT], w11[NVOX][VT];
    int tx[NVOX][VT], ty[NVOX][VT];
    unsigned valid = 0;
    {
        int bxmin[VT], bymin[VT], bxmax[VT], bymax[VT];
        const int big = 1 << 30;
#pragma unroll
        for (int v = 0; v < VT; ++v) { bxmin[v] = big; bymin[v] = big; bxmax[v] = -big; bymax[v] = -big; }
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            const int vx = kx * BXK + (col & 3) + kBX * u;
            vox[u] = (unsigned)(((long long)vx * Y + vy) * Z + vz);             // N < 2^28 (brick_fwd_supported)
            float c0, c1, c2;
            voxel_xyz(coords, b, N, vox[u], c0, c1, c2);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
                w00[u][v] = t.w00; w01[u][v] = t.w01; w10[u][v] = t.w10; w11[u][v] = t.w11;
                tx[u][v] = t.rx0; ty[u][v] = t.ry0;
                if (t.any) {
                    valid |= 1u << (u * VT + v);
                    bxmin[v] = t.rx0 < bxmin[v] ? t.rx0 : bxmin[v]; bxmax[v] = t.rx0 > bxmax[v] ? t.rx0 : bxmax[v];
                    bymin[v] = t.ry0 < bymin[v] ? t.ry0 : bymin[v]; bymax[v] = t.ry0 > bymax[v] ? t.ry0 : bymax[v];
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int nxmin = wave_max_dpp(-bxmin[v]), nymin = wave_max_dpp(-bymin[v]);
            const int xmax = wave_max_dpp(bxmax[v]), ymax = wave_max_dpp(bymax[v]);
            if (lane == 0 && xmax >= -nxmin) {
                atomicMin(&sh->bbox[v][0], -nxmin); atomicMin(&sh->bbox[v][1], -nymin);
                atomicMax(&sh->bbox[v][2], xmax); atomicMax(&sh->bbox[v][3], ymax);
            }
        }
    }
    EXP_FT(0);                                                                   // projections staged, tap records, wave boxes
    __syncthreads();
    EXP_FT(1);                                                                   // barrier: block boxes complete

    // ---- window per view (block-uniform): origin, column stride (odd) in slots, first slot; views packed back to back
    int wx0[VT], wy0[VT], ws[VT], nch[VT + 1], slot0[VT];
    nch[0] = 0;
    int used = 0, max_stride = 0;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
        const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
        int bw = 0, bh = 0;
        if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }     // taps reach x0+1, y0+1
        const int stride = bh | 1;
        const int chunks = (stride * bw + 63) >> 6;                              // 64-slot DMA chunks
        wx0[v] = xmin; wy0[v] = ymin; ws[v] = stride;
        max_stride = stride > max_stride ? stride : max_stride;
        slot0[v] = used;
        used += chunks << 6;
        nch[v + 1] = nch[v] + chunks;
    }
    // ring depth: 3 buffers (a DMA has between one and two iterations to land) when the windows fit a third of the pool, else 2
    const int cap3 = fwd_cap3(lds_slots), cap2 = fwd_cap2(lds_slots);
    const int nb = used <= cap3 ? 3 : 2;
    const int cap = nb == 3 ? cap3 : cap2;
    const int buf_bytes = kZeroBytes + cap * 16;
    const bool fits = used <= cap && nch[VT] <= MC * NW && max_stride + 2 <= kZeroSlots;
    TO *const obase = out + (long long)b * C * N;
    const float4 *const fk = featK + (long long)b * VT * nq * HW;

    if (fits) {
        for (int i = tid; i < kZeroSlots * nb; i += NT)
            *reinterpret_cast<float4 *>(smem + (i / kZeroSlots) * buf_bytes + (i % kZeroSlots) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- LDS byte offset (inside a buffer) of the nw tap; sw is 16 B further, ne one column stride further.  A sample that
        // is identically zero reads the zero region at the head of the buffer (long enough for "one stride further").
        int a0[NVOX][VT], ws16[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            ws16[v] = ws[v] * 16;
#pragma unroll
            for (int u = 0; u < NVOX; ++u) {
                const bool ok = (valid >> (u * VT + v)) & 1u;
                const int s0 = slot0[v] + (tx[u][v] - wx0[v]) * ws[v] + (ty[u][v] - wy0[v]);
                a0[u][v] = ok ? kZeroBytes + s0 * 16 : 0;
                if constexpr (kExp & 1) a0[u][v] = kZeroBytes + (lane + 64 * v + 256 * u) * 16;
            }
        }
        // ---- DMA chunks of this wave: chunk c covers 64 consecutive slots of one view's window
        unsigned g_off[MC];
        int l_dst[MC];
        int n_c = 0;
#pragma unroll
        for (int rr = 0; rr < MC; ++rr) {
            const int c = wave + rr * NW;
            l_dst[rr] = -1;
            g_off[rr] = 0;
            if (c < nch[VT]) {
                int v = 0;
#pragma unroll
                for (int uu = 1; uu < VT; ++uu) v += c >= nch[uu] ? 1 : 0;
                int sv = ws[0], ox = wx0[0], oy = wy0[0], c0 = nch[0], s0 = slot0[0];
#pragma unroll
                for (int uu = 1; uu < VT; ++uu) if (v == uu) { sv = ws[uu]; ox = wx0[uu]; oy = wy0[uu]; c0 = nch[uu]; s0 = slot0[uu]; }
                const int jj = c - c0, slot = (jj << 6) + lane;
                const int px = slot / sv, py = slot - px * sv;
                int gx = ox + px, gy = oy + py;                                  // pad row / columns past the window / outside the
                gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);                     // image: clamp -- those slots only meet zero weights
                gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
                g_off[rr] = (unsigned)((v * nq) * HW + gx * H + gy) * 16u;
                l_dst[rr] = kZeroBytes + (s0 + (jj << 6)) * 16;
                ++n_c;
            }
        }
        const unsigned lds_base = (unsigned)(size_t)(lds_void_t *)smem;
        auto ring = [&](int q) __attribute__((always_inline)) { return (nb == 3 ? q % 3 : q & 1) * buf_bytes; };
        auto dma = [&](int q) __attribute__((always_inline)) {
            const float4 *src = fk + (long long)q * HW;
            const int boff = ring(q);
#pragma unroll
            for (int rr = 0; rr < MC; ++rr)
                if (l_dst[rr] >= 0 && !(kExp & 4)) {
                    if constexpr (kExp & 512) glds16_m0(src, g_off[rr], lds_base + (unsigned)uniform(l_dst[rr] + boff));
                    else glds16(src, g_off[rr], lds_base + (unsigned)uniform(l_dst[rr] + boff));
                }
        };

        // ---- stores: lane 4a+b (+16h+32g) writes channel a, z = 16g + 4b .. 4b+3 of its column
        constexpr unsigned OSZ = sizeof(TO);
        const unsigned chan_bytes = (unsigned)(N * OSZ);
        unsigned st_off[NVOX];
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            const int z0 = ((lane >> 5) << 4) + ((lane & 3) << 2);
            st_off[u] = (vox[u] - (unsigned)zin + (unsigned)z0) * OSZ + (unsigned)((lane >> 2) & 3) * chan_bytes;
        }
        auto store_quad = [&](int q, int u, float (&res)[4]) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + (long long)(q * 4) * N, 0, (int)(4u * chan_bytes), 0x00020000);
            if constexpr (!(kExp & 32)) stride4_transpose(res, lane);
            if constexpr (kExp & 8) {
                asm volatile("" :: "v"(res[0]), "v"(res[1]), "v"(res[2]), "v"(res[3]), "s"(rs));
            } else if constexpr (OSZ == 4) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 d = {__builtin_bit_cast(unsigned, res[0]), __builtin_bit_cast(unsigned, res[1]),
                                 __builtin_bit_cast(unsigned, res[2]), __builtin_bit_cast(unsigned, res[3])};
                __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)st_off[u], 0, 18);   // nt sc1: best of the five policies (r02 ablations)
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 d = {pack2<TO>(res[0], res[1]), pack2<TO>(res[2], res[3])};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, (int)st_off[u], 0, 18);
            }
        };

        // ---- taps: two register sets of 4 x b128 (nw, ne, sw, se), used alternately by consecutive views
        f32x4 T[2][4];
        auto read_view = [&](int q, int u, int v, int set) __attribute__((always_inline)) {
            const int base = a0[u][v] + ring(q), far = base + ws16[v];
            if constexpr (kExp & 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(T[set][t].v[i]) : "v"(base), "v"(far));
                return;
            }
            T[set][0] = lds_tap(smem, base); T[set][2] = lds_tap(smem, base + 16);
            T[set][1] = lds_tap(smem, far); T[set][3] = lds_tap(smem, far + 16);
        };
        float sq[4][VT], sp[4][VT], res[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { T[0][t] = f32x4{{0.f, 0.f, 0.f, 0.f}}; T[1][t] = f32x4{{0.f, 0.f, 0.f, 0.f}}; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            res[i] = 0.f;
#pragma unroll
            for (int v = 0; v < VT; ++v) { sq[i][v] = 0.f; sp[i][v] = 0.f; }
        }
        for (int q = 0; q < nb - 1 && q < nq; ++q) dma(q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the zero regions are written
        EXP_FT(2);                                                               // windows, addresses, chunk table, first DMA issued
        const int a_step = wave & (VT - 1);
        // counted wait at the top of quad q: the DMA of quad q must have landed; at least NVOX stores (and, with three buffers,
        // the n_c DMAs of quad q+1) of this wave are younger.  The first and the last iterations lack part of that order.
        const int ncw = nb == 2 ? NVOX : NVOX + n_c;

        // PAR: which of sq / sp receives the samples of job u = 0 (alternates per quad when NVOX is odd)
        auto quad_iter = [&](int q, auto par_tag) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
            if constexpr (!(kExp & 256)) { if (q < 2 || q + 1 >= nq) wait_vmcnt(0); else wait_vmcnt(ncw); }
            if constexpr (!(kExp & 64)) bare_barrier();
#if MVHMR_EXP & 1024
            if (q == 0) EXP_FT(3);                                               // window 0 landed + barrier
#endif
            if (nb == 2 && q + 1 < nq) dma(q + 1);
#pragma unroll
            for (int u = 0; u < NVOX; ++u) {
                // this job's samples go to sq / sp alternately; the previous job's are aggregated in two halves between the folds
                // (the live samples stay 16 registers: rows of `prev` die as columns of `cur` are born)
                auto &cur = ((u + PAR) & 1) ? sp : sq;
                auto &prev = ((u + PAR) & 1) ? sq : sp;
                const bool st = q > 0 || u > 0;
                read_view(q, u, 0, 0);
                if constexpr (VT > 1) read_view(q, u, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                fwd_aggregate2<METHOD, VT>(prev[0], prev[1], res[0], res[1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < VT; ++v) {
                    if (v == (VT + 1) / 2) {
                        fwd_aggregate2<METHOD, VT>(prev[2], prev[3], res[2], res[3]);
                        if (st) store_quad(u > 0 ? q : q - 1, u > 0 ? u - 1 : NVOX - 1, res);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        cur[i][v] = bilerp(T[v & 1][0].v[i], T[v & 1][1].v[i], T[v & 1][2].v[i], T[v & 1][3].v[i], w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                        asm volatile("" : "+v"(cur[i][v]));                       // fold HERE: keeps the tap registers short-lived
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (v + 2 < VT) read_view(q, u, v + 2, v & 1);
                    if (nb == 3 && u == 0 && a_step == v && q + 2 < nq) dma(q + 2);   // the block's waves spread their DMAs over the job
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if constexpr (NVOX & 1) {
            for (int q = 0; q < nq; q += 2) {
                quad_iter(q, std::integral_constant<int, 0>{});
                if (q + 1 < nq) quad_iter(q + 1, std::integral_constant<int, 1>{});
            }
        } else {
            for (int q = 0; q < nq; ++q) quad_iter(q, std::integral_constant<int, 0>{});
        }
        // the last job: (nq - 1, NVOX - 1)
        const bool last_in_sp = ((NVOX & 1) ? nq - 1 : NVOX - 1) & 1;
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            if (last_in_sp) fwd_aggregate2<METHOD, VT>(sp[c], sp[c + 1], res[c], res[c + 1]);
            else fwd_aggregate2<METHOD, VT>(sq[c], sq[c + 1], res[c], res[c + 1]);
        }
        store_quad(nq - 1, NVOX - 1, res);
        EXP_FT(4);                                                               // the quad loop
#if MVHMR_EXP & 1024
        if (lane == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(&g_exp_fwd_timers[i], t_acc[i]);
            atomicAdd(&g_exp_fwd_timers[6], 1ull);
        }
#endif
    } else {
        // ---- windows do not fit the LDS pool: sample straight from global memory (clamped taps, zero weights outside)
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const int x0 = tx[u][v] < 0 ? 0 : tx[u][v], y0 = ty[u][v] < 0 ? 0 : ty[u][v];
                const int x1 = tx[u][v] + 1 > W - 1 ? W - 1 : tx[u][v] + 1, y1 = ty[u][v] + 1 > H - 1 ? H - 1 : ty[u][v] + 1;
                const int base = (v * nq) * HW;
                o00[v] = base + x0 * H + y0; o01[v] = base + x1 * H + y0; o10[v] = base + x0 * H + y1; o11[v] = base + x1 * H + y1;
            }
            for (int q = 0; q < nq; ++q) {
                const float4 *src = fk + (long long)q * HW;
                float s[4][VT];
#pragma unroll
                for (int v = 0; v < VT; ++v) {
                    const float4 a = src[o00[v]], bb = src[o01[v]], c = src[o10[v]], d = src[o11[v]];
                    s[0][v] = bilerp(a.x, bb.x, c.x, d.x, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    s[1][v] = bilerp(a.y, bb.y, c.y, d.y, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    s[2][v] = bilerp(a.z, bb.z, c.z, d.z, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    s[3][v] = bilerp(a.w, bb.w, c.w, d.w, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                }
                TO *oq = obase + (long long)(q * 4) * N;
#pragma unroll
                for (int i = 0; i < 4; ++i) (oq + i * N)[vox[u]] = from_f32<TO>(aggregate<METHOD, VT>(s[i]));
            }
        }
    }
}

// ---- launch of one instantiation (shared by the per-method translation units)
int fwd_lds_slots();                                  // 16-B LDS slots of the ring (one block per CU owns all 160 KiB)

template <int METHOD, int VT, int NT, typename TO, int NVOX>
hipError_t launch_fwd_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s)
{
    const int nbx = p.X / (kBX * NVOX), nby = p.Y / (NT / 128), nbz = p.Z / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int slots = fwd_lds_slots();
    const size_t lds = (size_t)slots * 16 + sizeof(FwdShared<VT>);
    auto kern = k_fwd_brick<METHOD, VT, NT, TO, NVOX>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int grid = ((nbx + tiles_x - 1) / tiles_x) * ((nby + tiles_y - 1) / tiles_y) * nbz * 8 * p.B;   // tile work items x 8 XCDs x samples
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, slots, total, make_gate(p, true));
    return hipGetLastError();
}

// 8 views in two groups of four, 1024 threads (brick_fwd_groups.h)
bool brick_fwd_grouped(const Problem &p);
template <int METHOD, int VT, typename TO>
hipError_t launch_fwd_groups_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s);

// one aggregation method: views x storage type x voxels per lane
template <int METHOD>
hipError_t launch_fwd_method(const void *featK_, const float *proj, const Coords &coords, void *out, const Problem &p, int nvox, hipStream_t s)
{
    const float4 *featK = static_cast<const float4 *>(featK_);
#define MVHMR_FWD_CASE(NVIEWS, NTHR, NV)                                                                                                 \
    if (p.V == NVIEWS && nvox == NV)                                                                                                      \
        return p.out_f16    ? launch_fwd_instance<METHOD, NVIEWS, NTHR, __half, NV>(featK, proj, coords, (__half *)out, p, s)            \
               : p.out_bf16 ? launch_fwd_instance<METHOD, NVIEWS, NTHR, bf16_t, NV>(featK, proj, coords, (bf16_t *)out, p, s)            \
                            : launch_fwd_instance<METHOD, NVIEWS, NTHR, float, NV>(featK, proj, coords, (float *)out, p, s)
    if (brick_fwd_grouped(p))
        return p.out_f16    ? launch_fwd_groups_instance<METHOD, 8, __half>(featK, proj, coords, (__half *)out, p, s)
               : p.out_bf16 ? launch_fwd_groups_instance<METHOD, 8, bf16_t>(featK, proj, coords, (bf16_t *)out, p, s)
                            : launch_fwd_groups_instance<METHOD, 8, float>(featK, proj, coords, (float *)out, p, s);
    MVHMR_FWD_CASE(2, 1024, 1); MVHMR_FWD_CASE(2, 1024, 2);
    MVHMR_FWD_CASE(4, 1024, 1); MVHMR_FWD_CASE(4, 1024, 2);
    MVHMR_FWD_CASE(8, 512, 1);  MVHMR_FWD_CASE(8, 512, 2);
#undef MVHMR_FWD_CASE
    return hipErrorNotSupported;
}

}  // namespace mvhmr
