// "plane" backward: gradient w.r.t. the feature maps for geometries the brick backward does not serve -- coarse grids, where
// neighbouring voxels share no taps and an LDS window has nothing to sum (BASELINE configs[1]: 32^3 at 2.9 px per voxel; the
// reference's shipped VOLUME_SIZE = 16, cfg/defaults.py:25, at 5.8).  The gather backward scatters every tap with a global float
// atomic there (4 per voxel, view and channel: atomic-rate bound, ~1.3 TB/s of added bytes on gfx950).
//
// Here the OUTPUT is what stays on chip, in two passes over a tap table:
//
//   tap table   k_plane_taps: per (sample, view, voxel) the four bilinear weights and the clamped tap coordinates, computed ONCE
//               (make_taps: the pinned projection arithmetic with its IEEE divides); the same kernel counts the taps every pixel
//               receives (the fixed-point headroom, below);
//   Jacobian    k_plane_ds: one thread per (voxel, channel quad) gathers the taps of all V views once, applies the aggregate's Jacobian
//               (autograd of models/aggregation.py:55-83) and writes ds = d(aggregate)/d(sample) * grad_out as one float4 per view into
//               a scratch stream dsW[b][v][quad][voxel] (V x the size of grad_out);
//   planes      k_bwd_plane: one block owns the gradient plane of one (sample, view, channel quad) -- Hf x Wf pixels x 4 channels,
//               147 KB for 96 x 96 maps -- in LDS, walks its ds stream and its view's table entries (36 B per voxel, coalesced), adds
//               the four taps and finally writes the plane once with plain coalesced stores, straight into the caller's planar
//               gradient tensor: no global atomic, no accumulator, no clear, no gradient layout pass.
//   (Round 3's first form did the Jacobian inside the plane blocks: every view's block re-sampled all V views, V-fold the gathers,
//   and the vector L1 was what bound it -- 1.48 ms for configs[1] against 0.3 + 0.2 ms for the two passes.)
//
//   plane       planar per channel, row stride Wf | 1 (a z column's taps are Wf-strided rows: an odd stride spreads them over the
//               banks), FIXED POINT: ds_add_f32 costs ~190 cycles per wave instruction on gfx950, ds_add_u32 4-6, ds_add_u64 ~8.
//               One scale per (view, channel), fixed BEFORE the walk from the exact max |ds| of that view's stream (the Jacobian pass
//               leaves it), so the walk needs neither a block-wide reduction nor a barrier -- the waves drift apart.  Three forms:
//                 int32 x 4 channels   the contributions are scaled to (2^31 - 2^23) / (multiplicity * max |ds|): no sum can overflow;
//                                      resolution 2^-31 * multiplicity of the view's largest |ds| per contribution.  Planes whose most
//                                      hit pixel receives <= kWideTaps taps;
//                 int64 x 2 channels   two walks over the stream, two channels each, in the same LDS bytes: contributions scaled to
//                                      the full int32 range, SUMS kept in 64 bits -- the resolution (2^-31 of the view's largest |ds|)
//                                      no longer depends on how many taps meet in a pixel.  Planes above kWideTaps (a far or zoomed-out
//                                      camera, a fine grid over a coarse map), and bounds that overflow fp32;
//                 int64 x 4 channels   the same in one walk, for maps small enough for 8-B cells (<= ~68 x 68): always;
//   non-finite  an Inf / NaN contribution cannot be carried in fixed point: its pixels are marked in a bit plane and written as NaN
//               -- exactly the pixels the reference's float scatter poisons (the gather variant's contract).
// Features are read from the column-major quad-planar fp32 copy (MVHMR_LAYOUT_QUAD: a tap = one 16-B load of 4 channels).
#include "brick_common.h"
#include "kernels.h"

namespace mvhmr {

namespace {

constexpr int plane_threads(int views) { return views == 8 ? 512 : 1024; }   // 8 views: 32 + 32 live samples and Jacobian terms need > 128 registers
constexpr int kPlaneLdsBytes = 160 * 1024 - 512;

struct PlaneShared {
    int gmax[4];         // max |ds| bits per channel of the block's stream (finite values only)
};

__host__ __device__ inline int plane_row_stride(int W) { return W | 1; }
__host__ __device__ inline size_t plane_lds_bytes(int H, int W, int cell_bytes = 4, int copies = 1)
{
    const size_t cells = (size_t)H * plane_row_stride(W);
    return 4 * cells * cell_bytes * copies + 4 * ((cells + 31) / 32) * sizeof(int) + sizeof(PlaneShared);
}
// maps small enough for 8-byte cells of all four channels
__host__ __device__ inline bool plane_wide4(int H, int W) { return plane_lds_bytes(H, W, 8) <= (size_t)kPlaneLdsBytes; }

// ------------------------------------------------------------------------------------------------- tap table
// One block per (sample, view): weights + packed clamped tap coordinates of every voxel, and the most taps any pixel of the plane
// receives -- counted in LDS (scattered global atomics run at 64 B per lane at the memory side: 0.13 ms for configs[1]).
__global__ void __launch_bounds__(1024)
k_plane_taps(const float *__restrict__ proj, const Coords coords, float4 *__restrict__ tabW, int *__restrict__ tabX,
             int *__restrict__ cmax, int V, int H, int W, long long N, Gate gate)
{
    if (gated_off(gate)) return;
    extern __shared__ int cnt[];                                                  // [H * W] + 1
    const int b = blockIdx.x / V, v = blockIdx.x % V, HW = H * W;
    for (int i = threadIdx.x; i <= HW; i += 1024) cnt[i] = 0;
    __shared__ float P[12];
    if (threadIdx.x < 12) P[threadIdx.x] = proj[((long long)b * V + v) * 12 + threadIdx.x];
    __syncthreads();
    const long long base = ((long long)b * V + v) * N;
    for (long long n = threadIdx.x; n < N; n += 1024) {
        float c0, c1, c2;
        voxel_xyz(coords, b, N, n, c0, c1, c2);
        const Taps t = make_taps(P, c0, c1, c2, H, W);
        tabW[base + n] = make_float4(t.w00, t.w01, t.w10, t.w11);
        tabX[base + n] = t.x0 | ((t.x1 - t.x0) << 15) | (t.y0 << 16) | ((t.y1 - t.y0) << 31);   // x0, y0 < 2^15; x1 - x0, y1 - y0 in {0, 1}
        if (t.w00 != 0.f) lds_add(cnt + t.y0 * W + t.x0, 1);
        if (t.w01 != 0.f) lds_add(cnt + t.y0 * W + t.x1, 1);
        if (t.w10 != 0.f) lds_add(cnt + t.y1 * W + t.x0, 1);
        if (t.w11 != 0.f) lds_add(cnt + t.y1 * W + t.x1, 1);
    }
    __syncthreads();
    int m = 1;
    for (int i = threadIdx.x; i < HW; i += 1024) { const int c = cnt[i]; m = c > m ? c : m; }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(&cnt[HW], m);
    __syncthreads();
    if (threadIdx.x == 0) cmax[blockIdx.x] = cnt[HW];
}

// ------------------------------------------------------------------------------------------------- the Jacobian pass
// ds = d(aggregate) / d(sample) * grad_out for every (voxel, view, channel), computed ONCE per (voxel, channel quad): one thread per
// voxel gathers the taps of all V views (16-B loads from the quad-planar copy), applies the aggregate's Jacobian and writes one float4
// (the quad's four channels) per view into dsW[b][v][quad][voxel] -- consecutive threads, consecutive voxels: 1-KiB stores.  No LDS,
// no barrier.  (The first form of the plane backward re-sampled all V views inside every view's plane block: V-fold the gathers, and
// the vector L1 -- TCP_GATE_EN1 99 % -- was what bound it: profiles/r03_plane_bwd_counters.txt.)
constexpr int kDsThreads = 256;
inline int ds_blocks_of(const Problem &p) { return (int)((p.N + kDsThreads - 1) / kDsThreads); }

template <int METHOD, int VT, typename TO>
__global__ void __launch_bounds__(kDsThreads)
k_plane_ds(const float4 *__restrict__ featK, const TO *__restrict__ grad_out, const float4 *__restrict__ tabW, const int *__restrict__ tabX,
           float4 *__restrict__ dsW, int *__restrict__ dsMax, int C, int H, int W, long long N, int q0, int nqs, Gate gate)
{
    if (gated_off(gate)) return;
    // the stream and the maxima hold ONE slab of nqs channel quads (q0 .. q0 + nqs - 1) at a time: indices qs = q - q0 inside them
    const int nq = C >> 2, HW = H * W;
    const int qs = blockIdx.y, q = q0 + qs, b = blockIdx.z;
    const bool live = (long long)blockIdx.x * kDsThreads + threadIdx.x < N;
    const long long n = live ? (long long)blockIdx.x * kDsThreads + threadIdx.x : N - 1;    // tail threads redo the last voxel, store nothing
    const float4 *const fq = featK + ((long long)b * VT * nq + q) * HW;          // view v: + v * nq * HW
    const TO *const gq = grad_out + ((long long)b * C + 4 * q) * N;
    const float4 *const tw = tabW + (long long)b * VT * N;
    const int *const tx = tabX + (long long)b * VT * N;
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = to_f32<TO>(gq[(long long)i * N + n]);
    float s[4][VT];
    if constexpr (METHOD == AGG_SOFTMAX || METHOD == AGG_MAX) {                    // sum / mean: the Jacobian does not depend on the samples
        // taps of two views in flight at a time (32 registers).  An identically zero sample reads one dummy pixel (0, 0) with zero
        // weights (the loads stay unconditional); a non-finite pixel there must not leak, hence the select in the fold
        constexpr int G = 2;
        f32x4 T[2][G][4];
        float4 wv[2][G];
        auto gather = [&](int v0, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int v = v0 + u;
                if (v >= VT) continue;
                wv[set][u] = tw[(long long)v * N + n];
                const unsigned xy = (unsigned)tx[(long long)v * N + n];
                const int x0 = xy & 0x7fff, x1 = x0 + ((xy >> 15) & 1), y0 = (xy >> 16) & 0x7fff, y1 = y0 + (xy >> 31);
                const float4 *fv = fq + (long long)v * nq * HW;
                const float4 a = fv[x0 * H + y0], bb = fv[x1 * H + y0], c = fv[x0 * H + y1], d = fv[x1 * H + y1];
                T[set][u][0] = f32x4{{a.x, a.y, a.z, a.w}}; T[set][u][1] = f32x4{{bb.x, bb.y, bb.z, bb.w}};
                T[set][u][2] = f32x4{{c.x, c.y, c.z, c.w}}; T[set][u][3] = f32x4{{d.x, d.y, d.z, d.w}};
            }
        };
        gather(0, 0);
#pragma unroll
        for (int v0 = 0; v0 < VT; v0 += G) {
            const int set = (v0 / G) & 1;
            if (v0 + G < VT) gather(v0 + G, set ^ 1);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int v = v0 + u;
                if (v >= VT) continue;
                const float4 w = wv[set][u];
                const bool zero = w.x == 0.f && w.y == 0.f && w.z == 0.f && w.w == 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = bilerp(T[set][u][0].v[i], T[set][u][1].v[i], T[set][u][2].v[i], T[set][u][3].v[i], w.x, w.y, w.z, w.w);
                    s[i][v] = zero ? 0.f : r;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < VT; ++v) s[i][v] = 0.f;
    }
    float ds[4][VT];
#pragma unroll
    for (int i = 0; i < 4; ++i) aggregate_grad<METHOD, VT>(s[i], g[i], ds[i]);
    float4 *const dst = dsW + ((long long)b * VT * nqs + qs) * N + n;            // view v: + v * nqs * N
    if (live) {
#pragma unroll
        for (int v = 0; v < VT; ++v) dst[(long long)v * nqs * N] = make_float4(ds[0][v], ds[1][v], ds[2][v], ds[3][v]);   // plain store: the plane kernel reads it next
    }
    // ---- max |ds| per (view, channel) over this block's voxels (finite values only: the plane kernel marks the others), for the
    // fixed-point scales -- per VIEW: a view whose gradients are small next to another view's keeps its own resolution.  Non-negative
    // floats order as their bit patterns.  One int4 per (block, view), no atomics: the plane blocks take the max over their sample's,
    // view's and quad's entries.
    __shared__ int wmax[kDsThreads / 64][VT][4];
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bits = __builtin_bit_cast(int, ds[i][v]) & 0x7fffffff;
            const int m = wave_max_to_last_row(bits < 0x7f800000 ? bits : 0);
            if ((threadIdx.x & 63) == 63) wmax[threadIdx.x >> 6][v][i] = m;
        }
    __syncthreads();
    if (threadIdx.x < VT * 4) {
        const int v = threadIdx.x >> 2, i = threadIdx.x & 3;
        int m = wmax[0][v][i];
#pragma unroll
        for (int w = 1; w < kDsThreads / 64; ++w) m = wmax[w][v][i] > m ? wmax[w][v][i] : m;
        dsMax[((((long long)b * nqs + qs) * VT + v) * gridDim.x + blockIdx.x) * 4 + i] = m;
    }
}

// ------------------------------------------------------------------------------------------------- the plane kernel
// One block per (sample, view, channel quad): reads its ds stream and its view's tap table (36 B per voxel, coalesced), adds the four
// taps of every voxel into the LDS plane, writes the plane once.  The fixed-point scales are fixed BEFORE the walk from the max |ds| of
// the sample's (view, channel) -- the Jacobian pass leaves one int4 per block and view -- and, in the int32 form, the plane's tap
// multiplicity.  WIDE4: 8-byte cells for all four channels (small maps); otherwise the block picks int32 x 4 channels or, for planes
// whose most hit pixel receives more than kWideTaps taps, int64 x 2 channels in two walks (the same LDS bytes).
constexpr int kWideTaps = 64;          // int32 form: resolution 2^-31 * taps of the view's max |ds| per contribution -> <= 3e-8 of it

__device__ __forceinline__ void lds_add64(long long *p, int v)
{
    __hip_atomic_fetch_add(p, (long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// COPIES (8-byte cells only): the block keeps that many private images of the four planes, one per 16-lane group of a wave.  Small
// maps are where many voxels share a pixel (the reference's shipped 16^3 volume over 12 x 12 maps: ~114 taps per pixel): the four
// columns a wave walks at a time land on the same few pixels, and same-address LDS atomics of one instruction are served one after
// the other.  With a copy per 16-lane group only the z-neighbours of one column still meet.
// NT: 1024 threads for planes that fill the LDS (one block per CU); 256 for small maps, whose blocks are short (a 16^3 volume is four
// iterations of 1024 threads): eight resident blocks per CU hide each other's load latency.
template <typename TF, bool WIDE4, int COPIES, int NT>
__global__ void __launch_bounds__(NT)
k_bwd_plane(const float4 *__restrict__ dsW, const int *__restrict__ dsMax, int ds_blocks, const float4 *__restrict__ tabW,
            const int *__restrict__ tabX, const int *__restrict__ cmax, TF *__restrict__ grad_features, int C, int V, int H, int W, long long N,
            int q0, int nqs, Gate gate)
{
    if (gated_off(gate)) return;
    constexpr int kPlaneThreads = NT;
    extern __shared__ __align__(16) unsigned char smem[];
    const int Ws = plane_row_stride(W), cells = H * Ws, mask_words = (cells + 31) >> 5;
    constexpr int CELL = WIDE4 ? 8 : 4;                                          // bytes per cell of the four-channel image
    int *const planes = reinterpret_cast<int *>(smem);                           // [4][cells] int32, [2][cells] / [4][cells] int64
    unsigned *const poison = reinterpret_cast<unsigned *>(smem + (size_t)4 * cells * CELL * COPIES);   // [4][mask_words]
    PlaneShared *const sh = reinterpret_cast<PlaneShared *>(poison + 4 * mask_words);
    const int tid = threadIdx.x;
    const int HW = H * W;
    const int qs = blockIdx.x % nqs, q = q0 + qs, v_own = (blockIdx.x / nqs) % V, b = blockIdx.x / (nqs * V);   // one slab of quads per launch
    const int plane_words = 4 * cells * (CELL / 4) * COPIES;
    const int copy_off = COPIES > 1 ? ((tid >> 4) & (COPIES - 1)) * 4 * cells : 0;     // this lane group's image (8-byte cells)

    for (int i = tid; i < plane_words + 4 * mask_words; i += kPlaneThreads) planes[i] = 0;    // the bit planes follow the planes
    if (tid < 4) sh->gmax[tid] = 0;
    __syncthreads();

    const float4 *const dq = dsW + (((long long)b * V + v_own) * nqs + qs) * N;
    const float4 *const tw = tabW + ((long long)b * V + v_own) * N;
    const int *const tx = tabX + ((long long)b * V + v_own) * N;
    // ---- max |ds| per channel over the sample's voxels, THIS view (the Jacobian pass's per-block maxima)
    {
        const int *const mq = dsMax + (((long long)b * nqs + qs) * V + v_own) * ds_blocks * 4;
        int gm = 0;
        for (int k = tid >> 2; k < ds_blocks; k += kPlaneThreads >> 2) { const int m = mq[k * 4 + (tid & 3)]; gm = m > gm ? m : gm; }
        if (gm) atomicMax(&sh->gmax[tid & 3], gm);
    }
    __syncthreads();
    const int cm = uniform(cmax[b * V + v_own]);
    float gm[4];
    bool int32_ok = !WIDE4 && cm <= kWideTaps;                                    // block-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gm[i] = __builtin_bit_cast(float, uniform(sh->gmax[i]));
        if (!((float)cm * gm[i] < 3.0e38f)) int32_ok = false;                     // the int32 bound itself overflows fp32: the wide form carries it
    }
    // scale per channel.  int32 form: every contribution is at most gmax (weights <= 1), at most `cm` of them meet in one pixel, so with
    // scale = (2^31 - 2^23) / (cm * gmax) no sum leaves int32 -- nothing of the 31 bits is given away to a power-of-two rounding.
    // int64 forms: scale = (2^31 - 2^23) / gmax, every contribution fits int32, the sums (< 2^31 taps) fit int64.
    float scale[4], inv_scale[4];                                                 // 0: every finite ds of this channel is zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float full = int32_ok ? (float)cm * gm[i] : gm[i];
        const bool ok = full > 0.f;
        scale[i] = uniform(ok ? __fdiv_rn(2139095040.f, full) : 0.f);            // wave-uniform: scalar registers
        inv_scale[i] = uniform(ok ? __fdiv_rn(full, 2139095040.f) : 0.f);
    }
    // a contribution fixed point cannot carry: Inf / NaN
    auto uncarried = [&](float dv) __attribute__((always_inline)) { return (__builtin_bit_cast(int, dv) & 0x7fffffff) >= 0x7f800000; };
    bool any_poison = false;
    TF *const out = grad_features + (((long long)b * V + v_own) * C + 4 * q) * HW;

    // one walk over the stream for channels [c_lo, c_hi); WIDE: 8-byte cells (plane index i - c_lo)
    auto walk = [&](auto wide_tag, int c_lo, int c_hi) __attribute__((always_inline)) {
        constexpr bool WIDE = decltype(wide_tag)::value;
        for (long long n0 = 0; n0 < N; n0 += kPlaneThreads) {
            const long long n = n0 + tid;
            float4 wo = make_float4(0.f, 0.f, 0.f, 0.f), dv = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned xo = 0;
            if (n < N) { wo = tw[n]; xo = (unsigned)tx[n]; dv = dq[n]; }
            const float d[4] = {dv.x, dv.y, dv.z, dv.w};
            // ---- this view's four taps into the plane (zero-weight taps -- outside the map, z <= 0 -- receive nothing: they add an
            // integer 0 to a clamped, valid pixel).  A non-finite contribution adds 0 here and marks its pixels below.
            const int x0 = xo & 0x7fff, x1 = x0 + ((xo >> 15) & 1), y0 = (xo >> 16) & 0x7fff, y1 = y0 + (xo >> 31);
            const int a00 = y0 * Ws + x0, a01 = y0 * Ws + x1, a10 = y1 * Ws + x0, a11 = y1 * Ws + x1;
            const bool taps = wo.x != 0.f || wo.y != 0.f || wo.z != 0.f || wo.w != 0.f;
            bool nf_any = false;
            if (taps) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < c_lo || i >= c_hi) continue;
                    const bool nf = uncarried(d[i]);
                    nf_any |= nf;
                    if (scale[i] == 0.f) continue;                                // block-uniform: nothing finite to add in this channel
                    const float dsc = nf ? 0.f : d[i] * scale[i];
                    if constexpr (WIDE) {
                        long long *pl = reinterpret_cast<long long *>(planes) + copy_off + (i - c_lo) * cells;
                        lds_add64(pl + a00, round_int(dsc * wo.x));
                        lds_add64(pl + a01, round_int(dsc * wo.y));
                        lds_add64(pl + a10, round_int(dsc * wo.z));
                        lds_add64(pl + a11, round_int(dsc * wo.w));
                    } else {
                        int *pl = planes + i * cells;
                        lds_add(pl + a00, round_int(dsc * wo.x));
                        lds_add(pl + a01, round_int(dsc * wo.y));
                        lds_add(pl + a10, round_int(dsc * wo.z));
                        lds_add(pl + a11, round_int(dsc * wo.w));
                    }
                }
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(nf_any) != 0, 0)) { // rare: exactly the pixels a float scatter would poison
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < c_lo || i >= c_hi || !taps || !uncarried(d[i])) continue;
                    unsigned *pm = poison + i * mask_words;
                    if (wo.x != 0.f) atomicOr(pm + (a00 >> 5), 1u << (a00 & 31));
                    if (wo.y != 0.f) atomicOr(pm + (a01 >> 5), 1u << (a01 & 31));
                    if (wo.z != 0.f) atomicOr(pm + (a10 >> 5), 1u << (a10 & 31));
                    if (wo.w != 0.f) atomicOr(pm + (a11 >> 5), 1u << (a11 & 31));
                    any_poison = true;
                }
            }
        }
    };
    // the plane, once, straight into the planar gradient tensor: channels [c_lo, c_hi)
    auto write_out = [&](auto wide_tag, int c_lo, int c_hi) __attribute__((always_inline)) {
        constexpr bool WIDE = decltype(wide_tag)::value;
        __syncthreads();
        const bool poisoned = __syncthreads_or(any_poison ? 1 : 0) != 0;
        for (int i = c_lo; i < c_hi; ++i) {
            const float inv = inv_scale[i];
            const unsigned *pm = poison + i * mask_words;
            for (int k = tid; k < HW; k += kPlaneThreads) {
                const int y = k / W, x = k - y * W, cell = y * Ws + x;
                float val;
                if constexpr (WIDE) {
                    const long long *pl = reinterpret_cast<const long long *>(planes) + (i - c_lo) * cells;
                    long long acc = pl[cell];
#pragma unroll
                    for (int k2 = 1; k2 < COPIES; ++k2) acc += pl[k2 * 4 * cells + cell];
                    val = (float)((double)acc * (double)inv);
                }
                else val = (float)(planes + i * cells)[cell] * inv;
                if (poisoned && ((pm[cell >> 5] >> (cell & 31)) & 1u)) val = __builtin_nanf("");
                out[(long long)i * HW + k] = from_f32<TF>(val);
            }
        }
    };
    if constexpr (WIDE4) {
        walk(std::true_type{}, 0, 4);
        write_out(std::true_type{}, 0, 4);
    } else if (int32_ok) {
        walk(std::false_type{}, 0, 4);
        write_out(std::false_type{}, 0, 4);
    } else {
        walk(std::true_type{}, 0, 2);
        write_out(std::true_type{}, 0, 2);
        __syncthreads();
        for (int i = tid; i < plane_words; i += kPlaneThreads) planes[i] = 0;
        __syncthreads();
        walk(std::true_type{}, 2, 4);
        write_out(std::true_type{}, 2, 4);
    }
}

template <int METHOD, int VT, typename TO>
hipError_t launch_ds_instance(const float4 *featK, const TO *grad_out, const float4 *tabW, const int *tabX, float4 *dsW, int *dsMax,
                              const Problem &p, int q0, int nqs, hipStream_t s)
{
    const dim3 grid((unsigned)ds_blocks_of(p), (unsigned)nqs, (unsigned)p.B);
    hipLaunchKernelGGL((k_plane_ds<METHOD, VT, TO>), grid, dim3(kDsThreads), 0, s, featK, grad_out, tabW, tabX, dsW, dsMax, p.C, p.H, p.W, p.N, q0, nqs,
                       make_gate(p, false));
    return hipGetLastError();
}

template <int METHOD, typename TO>
hipError_t launch_ds_views(const float4 *featK, const TO *go, const float4 *tabW, const int *tabX, float4 *dsW, int *dsMax, const Problem &p,
                           int q0, int nqs, hipStream_t s)
{
    switch (p.V) {
    case 2: return launch_ds_instance<METHOD, 2, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    case 4: return launch_ds_instance<METHOD, 4, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    case 8: return launch_ds_instance<METHOD, 8, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    }
    return hipErrorNotSupported;
}

template <typename TO>
hipError_t launch_ds_method(const float4 *featK, const TO *go, const float4 *tabW, const int *tabX, float4 *dsW, int *dsMax, const Problem &p,
                            int q0, int nqs, hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return launch_ds_views<AGG_SOFTMAX, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    case AGG_SUM: return launch_ds_views<AGG_SUM, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    case AGG_MEAN: return launch_ds_views<AGG_MEAN, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    case AGG_MAX: return launch_ds_views<AGG_MAX, TO>(featK, go, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
    }
    return hipErrorInvalidValue;
}

template <typename TF>
hipError_t launch_plane_instance(const float4 *dsW, const int *dsMax, const float4 *tabW, const int *tabX, const int *cmax, TF *grad_features,
                                 const Problem &p, int q0, int nqs, hipStream_t s)
{
    const bool wide4 = plane_wide4(p.H, p.W);
    const bool copies4 = wide4 && plane_lds_bytes(p.H, p.W, 8, 4) <= (size_t)kPlaneLdsBytes / 8;   // eight blocks per CU (maps up to ~16 x 16)
    const size_t lds = plane_lds_bytes(p.H, p.W, wide4 ? 8 : 4, copies4 ? 4 : 1);
    auto kern = copies4 ? k_bwd_plane<TF, true, 4, 256> : wide4 ? k_bwd_plane<TF, true, 1, 1024> : k_bwd_plane<TF, false, 1, 1024>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(p.B * p.V * nqs);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(copies4 ? 256 : 1024), lds, s, dsW, dsMax, ds_blocks_of(p), tabW, tabX, cmax, grad_features, p.C, p.V, p.H, p.W, p.N,
                       q0, nqs, make_gate(p, false));
    return hipGetLastError();
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

// ------------------------------------------------------------------------------------------------- host side
bool plane_bwd_supported(const Problem &p)
{
    if (p.V != 2 && p.V != 4 && p.V != 8) return false;
    if (p.C % 4 || p.W > 32767 || p.H > 32767) return false;
    if (plane_lds_bytes(p.H, p.W) > (size_t)kPlaneLdsBytes) return false;        // 96 x 96 maps: 149 KB
    if ((long long)p.B * p.V * (p.C / 4) * p.H * p.W >= (1ll << 31)) return false;   // 32-bit tap offsets into the staged copy
    if ((long long)p.B * p.V * (p.C / 4) >= (1ll << 31) || p.B > 65535) return false;
    if (p.N >= (1ll << 31) / 1024 * 1024 || p.C / 4 > 65535) return false;       // grid dimensions of the Jacobian pass
    // (a plane that sums hundreds of taps per pixel -- a fine grid over a coarse map, a far camera -- keeps its sums in 64 bits: the
    // resolution does not depend on the multiplicity.  Round 3 sent those to the float scatter above 32 taps per pixel on average.)
    if (plane_table_bytes(p) > ((size_t)4 << 30)) return false;                  // a fine grid forced onto this path: keep the old scatter
    return true;
}

// The Jacobian stream is V x the size of grad_out in fp32.  It is produced and consumed one SLAB of channel quads at a time (k_plane_ds then
// k_bwd_plane per slab), sized to ~256 MB, so that
// the scratch stays bounded (configs[1]: 0.27 GB instead of 1.07 GB, at +0.02 ms for the extra launches).  (A slab that fits the
// Infinity Cache, 128 MB, is not faster: 1.303 vs 1.28 ms.)
int plane_slab_quads(const Problem &p)
{
    const size_t per_quad = (size_t)p.B * p.V * (size_t)p.N * sizeof(float4);
    size_t n = ((size_t)256 << 20) / (per_quad ? per_quad : 1);
    if (n < 1) n = 1;
    if (n > (size_t)(p.C / 4)) n = (size_t)(p.C / 4);
    return (int)n;
}
// [ weights float4 (B,V,N) | packed tap coordinates int (B,V,N) | max tap count int (B,V) | ds float4 (B,V,slab,N) |
//   max |ds| int4 (B, slab, V, blocks of the Jacobian pass) ]
size_t plane_table_bytes(const Problem &p)
{
    const size_t bvn = (size_t)p.B * p.V * (size_t)p.N;
    const size_t slab = (size_t)plane_slab_quads(p);
    return align256(bvn * sizeof(float4)) + align256(bvn * sizeof(int)) + align256((size_t)p.B * p.V * sizeof(int)) +
           align256(bvn * slab * sizeof(float4)) + align256((size_t)p.B * slab * p.V * ds_blocks_of(p) * 4 * sizeof(int));
}

// featK: column-major quad-planar fp32 copy of the features; grad_features: the caller's PLANAR gradient tensor (B,V,C,Hf,Wf), every
// element written; table: plane_table_bytes(p) bytes of scratch
hipError_t launch_bwd_plane(const void *featK, const void *grad_out, const float *proj, const Coords &coords, void *grad_features,
                            void *table, const Problem &p, hipStream_t s)
{
    if (!plane_bwd_supported(p)) return hipErrorNotSupported;
    const size_t bvn = (size_t)p.B * p.V * (size_t)p.N;
    unsigned char *t = static_cast<unsigned char *>(table);
    float4 *tabW = reinterpret_cast<float4 *>(t);
    int *tabX = reinterpret_cast<int *>(t + align256(bvn * sizeof(float4)));
    int *cmax = reinterpret_cast<int *>(t + align256(bvn * sizeof(float4)) + align256(bvn * sizeof(int)));
    unsigned char *t3 = t + align256(bvn * sizeof(float4)) + align256(bvn * sizeof(int)) + align256((size_t)p.B * p.V * sizeof(int));
    const int slab = plane_slab_quads(p);
    float4 *dsW = reinterpret_cast<float4 *>(t3);
    int *dsMax = reinterpret_cast<int *>(t3 + align256(bvn * (size_t)slab * sizeof(float4)));
    const Gate gate = make_gate(p, false);
    hipLaunchKernelGGL(k_plane_taps, dim3((unsigned)(p.B * p.V)), dim3(1024), (size_t)(p.H * p.W + 1) * sizeof(int), s, proj, coords, tabW, tabX, cmax,
                       p.V, p.H, p.W, p.N, gate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const float4 *fk = static_cast<const float4 *>(featK);
    // per slab of quads: the Jacobian pass (grad_out's storage type), then the planes (the gradient's storage type)
    for (int q0 = 0; q0 < p.C / 4; q0 += slab) {
        const int nqs = p.C / 4 - q0 < slab ? p.C / 4 - q0 : slab;
        if (p.out_bf16) e = p.feat_f16 ? hipErrorNotSupported : launch_ds_method<bf16_t>(fk, (const bf16_t *)grad_out, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
        else if (p.out_f16) e = launch_ds_method<__half>(fk, (const __half *)grad_out, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
        else e = launch_ds_method<float>(fk, (const float *)grad_out, tabW, tabX, dsW, dsMax, p, q0, nqs, s);
        if (e != hipSuccess) return e;
        e = p.feat_f16 ? launch_plane_instance<__half>(dsW, dsMax, tabW, tabX, cmax, (__half *)grad_features, p, q0, nqs, s)
                       : launch_plane_instance<float>(dsW, dsMax, tabW, tabX, cmax, (float *)grad_features, p, q0, nqs, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mvhmr
