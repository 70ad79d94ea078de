// "plane" backward: gradient w.r.t. the feature maps for geometries the brick backward does not serve -- coarse grids, where
// neighbouring voxels share no taps and an LDS window has nothing to sum (BASELINE configs[1]: 32^3 at 2.9 px per voxel; the
// reference's shipped VOLUME_SIZE = 16, cfg/defaults.py:25, at 5.8).  The gather backward scatters every tap with a global float
// atomic there (4 per voxel, view and channel: atomic-rate bound, ~1.3 TB/s of added bytes on gfx950).
//
// Here the OUTPUT is what stays on chip: one block owns the gradient plane of one (sample, view, channel quad) -- Hf x Wf pixels x 4
// channels, 147 KB for 96 x 96 maps -- in LDS, walks every voxel of the sample, re-samples all V views for the aggregate's Jacobian
// (autograd of models/aggregation.py:55-83, V-fold redundant across the views' blocks), adds only ITS view's four taps into the plane
// and finally writes the plane once with plain coalesced stores, straight into the caller's planar gradient tensor: no global atomic,
// no accumulator, no clear, no gradient layout pass.
//
//   tap table   k_plane_taps: per (sample, view, voxel) the four bilinear weights and the clamped tap coordinates, computed ONCE
//               (make_taps: the pinned projection arithmetic with its IEEE divides) instead of by each of the V * C/4 blocks that
//               visit the voxel; the same kernel counts the taps every pixel receives (the fixed-point headroom, below);
//   plane       planar per channel, row stride Wf | 1 (a z column's taps are Wf-strided rows: an odd stride spreads them over the
//               banks), int32 FIXED POINT: ds_add_f32 costs ~190 cycles per wave instruction on gfx950, ds_add_u32 4-6.  One
//               power-of-two scale per channel, fixed BEFORE the walk from a bound on every contribution the block can meet:
//               |ds| <= max |grad_out| (this sample, this channel) * (1 + R), R = the range the samples of this quad can span
//               (max(0, feature max) - min(0, feature min) over the V planes: a bilinear sample with zero padding is a sub-convex
//               combination of its taps; the softmax Jacobian is g p_v (1 + s_v - out), the others are <= |g|), times the plane's
//               tap multiplicity: no sum can overflow, and the walk needs neither a block-wide reduction nor a barrier -- the waves
//               drift apart and their gathers, arithmetic and LDS adds overlap.  Resolution: 2^-31 * multiplicity of that bound per
//               contribution (observed: <= 6e-6 of the largest gradient with ~100 taps per pixel and 2^40 of dynamic range);
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
    int gmax[4];         // max |grad_out| bits per channel of the quad over the sample (finite values only)
    int fpos, fneg;      // bits of max(0, feature max) and of max(0, -feature min) over the V planes of the quad (finite values only)
};

__host__ __device__ inline int plane_row_stride(int W) { return W | 1; }
__host__ __device__ inline size_t plane_lds_bytes(int H, int W)
{
    const size_t cells = (size_t)H * plane_row_stride(W);
    return 4 * cells * sizeof(int) + 4 * ((cells + 31) / 32) * sizeof(int) + sizeof(PlaneShared);
}

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

// ------------------------------------------------------------------------------------------------- the plane kernel
template <int METHOD, int VT, typename TO, typename TF>
__global__ void __launch_bounds__(plane_threads(VT))
k_bwd_plane(const float4 *__restrict__ featK, const TO *__restrict__ grad_out, const float4 *__restrict__ tabW,
            const int *__restrict__ tabX, const int *__restrict__ cmax, TF *__restrict__ grad_features, int C, int H, int W,
            long long N, Gate gate)
{
    if (gated_off(gate)) return;
    constexpr int kPlaneThreads = plane_threads(VT);
    extern __shared__ __align__(16) unsigned char smem[];
    const int Ws = plane_row_stride(W), cells = H * Ws, mask_words = (cells + 31) >> 5;
    int *const planes = reinterpret_cast<int *>(smem);                           // [4][cells]
    unsigned *const poison = reinterpret_cast<unsigned *>(planes + 4 * cells);   // [4][mask_words]
    PlaneShared *const sh = reinterpret_cast<PlaneShared *>(poison + 4 * mask_words);
    const int tid = threadIdx.x, lane = tid & 63;
    const int nq = C >> 2, HW = H * W;
    const int q = blockIdx.x % nq, v_own = (blockIdx.x / nq) % VT, b = blockIdx.x / (nq * VT);

    for (int i = tid; i < 4 * cells + 4 * mask_words; i += kPlaneThreads) planes[i] = 0;      // the bit planes follow the planes
    if (tid < 4) sh->gmax[tid] = 0;
    if (tid == 4) { sh->fpos = 0; sh->fneg = 0; }
    __syncthreads();

    const float4 *const fq = featK + ((long long)b * VT * nq + q) * HW;          // view v: + v * nq * HW
    const TO *const gq = grad_out + ((long long)b * C + 4 * q) * N;
    const float4 *const tw = tabW + (long long)b * VT * N;
    const int *const tx = tabX + (long long)b * VT * N;
    // ---- the bound that fixes the scales (finite values only: a non-finite contribution takes the bit-plane route below).
    // Non-negative floats order as their bit patterns, so the reductions run on ints.
    {
        int gm[4] = {0, 0, 0, 0}, fp = 0, fn = 0;
        for (long long n = tid; n < N; n += kPlaneThreads)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bits = __builtin_bit_cast(int, to_f32<TO>(gq[(long long)i * N + n])) & 0x7fffffff;
                gm[i] = bits < 0x7f800000 && bits > gm[i] ? bits : gm[i];
            }
        for (int v = 0; v < (METHOD == AGG_SOFTMAX ? VT : 0); ++v)               // only the softmax Jacobian depends on the samples
            for (int k = tid; k < HW; k += kPlaneThreads) {
                const float4 f = fq[(long long)v * nq * HW + k];
                const float e[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int raw = __builtin_bit_cast(int, e[i]), bits = raw & 0x7fffffff;
                    if (bits < 0x7f800000) { if (raw < 0) fn = bits > fn ? bits : fn; else fp = bits > fp ? bits : fp; }
                }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int m = wave_max_dpp(gm[i]); if (lane == 0 && m) atomicMax(&sh->gmax[i], m); }
        const int mp = wave_max_dpp(fp), mn = wave_max_dpp(fn);
        if (lane == 0) { if (mp) atomicMax(&sh->fpos, mp); if (mn) atomicMax(&sh->fneg, mn); }
    }
    __syncthreads();
    // scale per channel: every contribution is at most bound = gmax * (1 + R) (times 1 + 2^-8 for the rounding of the Jacobian's own
    // arithmetic), at most `cm` of them meet in one pixel, so with scale = (2^31 - 2^23) / (cm * bound) no sum leaves int32 -- and
    // nothing of the 31 bits is given away to a power-of-two rounding of the bound or of the multiplicity
    const int cm = uniform(cmax[b * VT + v_own]);
    const float range = METHOD == AGG_SOFTMAX ? __builtin_bit_cast(float, uniform(sh->fpos)) + __builtin_bit_cast(float, uniform(sh->fneg)) : 0.f;
    float scale[4], inv_scale[4];                                                 // 0: every finite grad_out of this channel is zero
    bool over[4];                                                                 // the bound itself overflows fp32: nothing can be scaled
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float full = (float)cm * (__builtin_bit_cast(float, uniform(sh->gmax[i])) * (1.f + range) * 1.00390625f);
        const bool ok = full > 0.f && full < 3.0e38f;
        over[i] = !(full < 3.0e38f);                                              // (inf included): non-zero contributions become NaN pixels
        scale[i] = uniform(ok ? __fdiv_rn(2139095040.f, full) : 0.f);            // wave-uniform: scalar registers
        inv_scale[i] = uniform(ok ? __fdiv_rn(full, 2139095040.f) : 0.f);
    }
    // a contribution fixed point cannot carry: Inf / NaN, or anything non-zero under an overflowing bound
    auto uncarried = [&](float dv, int i) __attribute__((always_inline)) {
        return (__builtin_bit_cast(int, dv) & 0x7fffffff) >= 0x7f800000 || (over[i] && dv != 0.f);
    };
    bool any_poison = false;

    // table entries and grad_out of the NEXT chunk are requested before this chunk's arithmetic: the taps' addresses come from the
    // table, so without the prefetch every chunk pays two dependent memory latencies back to back
    float4 wn[VT];
    int xn[VT];
    float gn[4];
    auto request = [&](long long n) __attribute__((always_inline)) {
        const bool in = n < N;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            wn[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            xn[v] = 0;
            if (in) { wn[v] = tw[(long long)v * N + n]; xn[v] = tx[(long long)v * N + n]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) gn[i] = in ? to_f32<TO>(gq[(long long)i * N + n]) : 0.f;
    };
    constexpr bool kPrefetch = VT <= 4;                                            // 8 views: the second set of entries (48 registers) would spill
    if constexpr (kPrefetch) request(tid);
    int chunk = 0;
    for (long long n0 = 0; n0 < N; n0 += kPlaneThreads, ++chunk) {
        if constexpr (!kPrefetch) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gn[i] = n0 + tid < N ? to_f32<TO>(gq[(long long)i * N + n0 + tid]) : 0.f;
        }
        float s[4][VT];
        float4 wo = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned xo = 0;
        // taps of two views in flight at a time (32 registers): the next pair is requested before this pair is folded.  An
        // identically zero sample reads one dummy pixel (0, 0) with zero weights (the loads stay unconditional); a non-finite pixel
        // there must not leak, hence the select in the fold
        constexpr int G = 2;
        f32x4 T[2][G][4];
        auto gather = [&](int v0, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int v = v0 + u;
                if (v >= VT) continue;
                if constexpr (!kPrefetch) {                                       // 8 views: a pair's entries right before its taps
                    wn[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                    xn[v] = 0;
                    if (n0 + tid < N) { wn[v] = tw[(long long)v * N + n0 + tid]; xn[v] = tx[(long long)v * N + n0 + tid]; }
                }
                const unsigned xy = (unsigned)xn[v];
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
                const float4 w = wn[v];
                if (v == v_own) { wo = w; xo = (unsigned)xn[v]; }
                const bool zero = w.x == 0.f && w.y == 0.f && w.z == 0.f && w.w == 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = bilerp(T[set][u][0].v[i], T[set][u][1].v[i], T[set][u][2].v[i], T[set][u][3].v[i], w.x, w.y, w.z, w.w);
                    s[i][v] = zero ? 0.f : r;
                }
            }
        }
        // d(aggregate)/d(sample of this block's view) * grad_out, per channel
        float d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float ds[VT];
            aggregate_grad<METHOD, VT>(s[i], gn[i], ds);
            float dv = ds[0];
#pragma unroll
            for (int v = 1; v < VT; ++v) dv = v == v_own ? ds[v] : dv;
            d[i] = dv;
        }
        if constexpr (kPrefetch) request(n0 + kPlaneThreads + tid);                // in flight under the adds
        // ---- this view's four taps into the plane (zero-weight taps -- outside the map, z <= 0 -- receive nothing: they add an
        // integer 0 to a clamped, valid pixel).  A non-finite contribution adds 0 here and marks its pixels below.
        const int x0 = xo & 0x7fff, x1 = x0 + ((xo >> 15) & 1), y0 = (xo >> 16) & 0x7fff, y1 = y0 + (xo >> 31);
        const int a00 = y0 * Ws + x0, a01 = y0 * Ws + x1, a10 = y1 * Ws + x0, a11 = y1 * Ws + x1;
        const bool taps = wo.x != 0.f || wo.y != 0.f || wo.z != 0.f || wo.w != 0.f;
        bool nf_any = false;
        if (taps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool nf = uncarried(d[i], i);
                nf_any |= nf;
                if (scale[i] == 0.f) continue;                                    // block-uniform: nothing finite to add in this channel
                int *pl = planes + i * cells;
                const float dsc = nf ? 0.f : d[i] * scale[i];
                lds_add(pl + a00, round_int(dsc * wo.x));
                lds_add(pl + a01, round_int(dsc * wo.y));
                lds_add(pl + a10, round_int(dsc * wo.z));
                lds_add(pl + a11, round_int(dsc * wo.w));
            }
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(nf_any) != 0, 0)) {     // rare: exactly the pixels a float scatter would poison
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!taps || !uncarried(d[i], i)) continue;
                unsigned *pm = poison + i * mask_words;
                if (wo.x != 0.f) atomicOr(pm + (a00 >> 5), 1u << (a00 & 31));
                if (wo.y != 0.f) atomicOr(pm + (a01 >> 5), 1u << (a01 & 31));
                if (wo.z != 0.f) atomicOr(pm + (a10 >> 5), 1u << (a10 & 31));
                if (wo.w != 0.f) atomicOr(pm + (a11 >> 5), 1u << (a11 & 31));
                any_poison = true;
            }
        }
    }
    __syncthreads();
    // ---- the plane, once, straight into the planar gradient tensor
    const bool poisoned = __syncthreads_or(any_poison ? 1 : 0) != 0;
    TF *const out = grad_features + (((long long)b * VT + v_own) * C + 4 * q) * HW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float inv = inv_scale[i];
        const int *pl = planes + i * cells;
        const unsigned *pm = poison + i * mask_words;
        for (int k = tid; k < HW; k += kPlaneThreads) {
            const int y = k / W, x = k - y * W, cell = y * Ws + x;
            float val = (float)pl[cell] * inv;
            if (poisoned && ((pm[cell >> 5] >> (cell & 31)) & 1u)) val = __builtin_nanf("");
            out[(long long)i * HW + k] = from_f32<TF>(val);
        }
    }
}

template <int METHOD, int VT, typename TO, typename TF>
hipError_t launch_plane_instance(const float4 *featK, const TO *grad_out, const float4 *tabW, const int *tabX, const int *cmax,
                                 TF *grad_features, const Problem &p, hipStream_t s)
{
    const size_t lds = plane_lds_bytes(p.H, p.W);
    auto kern = k_bwd_plane<METHOD, VT, TO, TF>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(p.B * VT * (p.C / 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(plane_threads(VT)), lds, s, featK, grad_out, tabW, tabX, cmax, grad_features, p.C, p.H, p.W, p.N,
                       make_gate(p, false));
    return hipGetLastError();
}

template <int METHOD, typename TO, typename TF>
hipError_t launch_plane_views(const float4 *featK, const TO *go, const float4 *tabW, const int *tabX, const int *cmax, TF *gf,
                              const Problem &p, hipStream_t s)
{
    switch (p.V) {
    case 2: return launch_plane_instance<METHOD, 2, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    case 4: return launch_plane_instance<METHOD, 4, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    case 8: return launch_plane_instance<METHOD, 8, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    }
    return hipErrorNotSupported;
}

template <typename TO, typename TF>
hipError_t launch_plane_method(const float4 *featK, const TO *go, const float4 *tabW, const int *tabX, const int *cmax, TF *gf,
                               const Problem &p, hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return launch_plane_views<AGG_SOFTMAX, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    case AGG_SUM: return launch_plane_views<AGG_SUM, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    case AGG_MEAN: return launch_plane_views<AGG_MEAN, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    case AGG_MAX: return launch_plane_views<AGG_MAX, TO, TF>(featK, go, tabW, tabX, cmax, gf, p, s);
    }
    return hipErrorInvalidValue;
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
    if (plane_table_bytes(p) > ((size_t)1 << 30)) return false;                  // a fine grid forced onto this path: keep the old scatter
    return true;
}

// [ weights float4 (B,V,N) | packed tap coordinates int (B,V,N) | max tap count int (B,V) ]
size_t plane_table_bytes(const Problem &p)
{
    const size_t bvn = (size_t)p.B * p.V * (size_t)p.N;
    return align256(bvn * sizeof(float4)) + align256(bvn * sizeof(int)) + align256((size_t)p.B * p.V * sizeof(int));
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
    const Gate gate = make_gate(p, false);
    hipLaunchKernelGGL(k_plane_taps, dim3((unsigned)(p.B * p.V)), dim3(1024), (size_t)(p.H * p.W + 1) * sizeof(int), s, proj, coords, tabW, tabX, cmax,
                       p.V, p.H, p.W, p.N, gate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const float4 *fk = static_cast<const float4 *>(featK);
    if (p.out_bf16) return p.feat_f16 ? hipErrorNotSupported : launch_plane_method<bf16_t, float>(fk, (const bf16_t *)grad_out, tabW, tabX, cmax, (float *)grad_features, p, s);
    if (!p.out_f16 && !p.feat_f16) return launch_plane_method<float, float>(fk, (const float *)grad_out, tabW, tabX, cmax, (float *)grad_features, p, s);
    if (p.out_f16 && p.feat_f16) return launch_plane_method<__half, __half>(fk, (const __half *)grad_out, tabW, tabX, cmax, (__half *)grad_features, p, s);
    if (!p.out_f16 && p.feat_f16) return launch_plane_method<float, __half>(fk, (const float *)grad_out, tabW, tabX, cmax, (__half *)grad_features, p, s);
    return hipErrorNotSupported;
}

}  // namespace mvhmr
