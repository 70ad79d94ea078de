// "brick" variant of the fused un-projection forward: LDS-staged feature patches.
//
// Why.  Per (voxel, view, channel) the sampler moves 4 taps x 4 B from the memory system into registers:
// 137 GB at the north-star size (64^3 x 256 ch x 4 views x 32 samples) against 9.9 GB of compulsory HBM
// traffic.  Through the vector L1 (64 B/clk/CU, ~17 TB/s chip-wide for gathers) that alone costs ~8 ms --
// the gather variant's time.  LDS delivers 256 B/clk/CU to ds_read_b128, so the taps must come from LDS,
// and every feature pixel must be fetched from L2 far less often than it is sampled.
//
// Mapping (CDNA4, wave64).
//   block  = one voxel brick of 4 x (NT/128 * VPL) x 32 voxels of one sample, NT threads; a lane owns VPL
//            fixed voxels for the block's lifetime, so its tap records (LDS address, 4 weights per view)
//            are computed ONCE (device_common.h::make_taps, reference aggregation.py:38-54) and stay in
//            registers while the block loops over all channel quads;
//   z-long bricks: the (B,C,X,Y,Z) output is written in runs of 32 consecutive z = 128 B per (channel,
//            column) -- measured floor for full-rate HBM writes on MI355X (64-B runs: 3.4 TB/s, 32-B: 0.7);
//   layout = features are re-laid "quad-planar" (B,V,C/4,Hf,Wf,4) by a pre-pass, so a pixel's 4 channels
//            are one 16-B LDS slot and a window row is one contiguous global segment;
//   window = per view, the bounding box of the brick's taps (wave shuffles + one LDS atomic per wave);
//            the views' windows are packed back to back in one LDS pool (a brick near one camera of the
//            ring is far from the opposite one, so the SUM of the windows is what has to fit);
//   staging = LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), per channel quad, double-buffered:
//            the DMA of quad q+1 is in flight while quad q is sampled; one raw s_barrier per quad behind a
//            counted s_waitcnt vmcnt(N) that leaves the output stores in flight;
//   bricks whose windows do not fit the pool (exotic cameras, huge maps) take a slower block-uniform path
//            that samples from global memory, so geometry can only cost speed, never correctness;
//   zero padding = a tap outside the image has weight 0 (make_taps) and its staged pixel is clamped into
//            the image, i.e. contributes 0 * finite (aggregation.py:55-58, padding_mode='zeros').
// The cross-view aggregate runs in registers exactly as in the gather variant (aggregation.py:71-85).
#include "device_common.h"
#include "kernels.h"

namespace mvhmr {

constexpr int kBZ = 32;            // z extent of a brick: 128-B output runs
constexpr int kBX = 4;
constexpr int kMaxChunks = 5;      // 64-slot DMA chunks per wave per quad

// workgroup barrier that waits for this wave's LDS operations only (not for global loads / stores in flight)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) void lds_void_t;

template <int VT>
struct BrickShared {
    int bbox[VT][4];               // xmin, ymin, xmax, ymax of the nw taps (valid voxels only)
    float proj[VT][12];
};

__device__ __forceinline__ int wave_min(int x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int y = __shfl_xor(x, m); x = y < x ? y : x; }
    return x;
}
__device__ __forceinline__ int wave_max(int x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int y = __shfl_xor(x, m); x = y > x ? y : x; }
    return x;
}

// features (BV, C, HW) fp32 -> (BV, C/4, HW, 4)
__global__ void __launch_bounds__(256)
k_to_quad_planar(const float *__restrict__ src, float4 *__restrict__ dst, int C, int HW)
{
    const long long bv = blockIdx.z;
    const int q = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float *s = src + (bv * C + q * 4) * HW + p;
    dst[(bv * (C >> 2) + q) * HW + p] = make_float4(s[0], s[HW], s[2 * (long long)HW], s[3 * (long long)HW]);
}

// NT threads, brick = 4 x (NT/128) x 32 voxels, one voxel per lane.
// LDS: [ buffer 0 | buffer 1 | BrickShared ], each buffer = 64 B of zeros + `cap` 16-B slots shared by the views.
template <int METHOD, int VT, int NT>
__global__ void __launch_bounds__(NT)
k_fwd_brick(const float4 *__restrict__ featK, const float *__restrict__ proj, const float *__restrict__ coords,
            float *__restrict__ out, int C, int H, int W, int X, int Y, int Z, int nby, int nbz, int bricks_per_sample,
            int cap, int total_blocks)
{
    constexpr int BY = NT / 128, NW = NT / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const int buf_bytes = 64 + cap * 16;
    BrickShared<VT> *sh = reinterpret_cast<BrickShared<VT> *>(smem + 2 * buf_bytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order: blocks b, b+8, ... share an XCD (round-robin dispatch); give each XCD a contiguous
    // range of bricks so that neighbouring windows meet in the same L2.  Placement only affects speed.
    const int per_xcd = (total_blocks + 7) >> 3;
    const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (work >= total_blocks) return;
    const int b = work / bricks_per_sample, brick = work % bricks_per_sample;
    const int kz = brick % nbz, ky = (brick / nbz) % nby, kx = brick / (nbz * nby);
    const long long N = (long long)X * Y * Z;
    const int HW = H * W, nq = C >> 2;

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = proj[((long long)b * VT) * 12 + tid];
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    if (tid < 8) *reinterpret_cast<float4 *>(smem + (tid >> 2) * buf_bytes + (tid & 3) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // ---- this lane's voxel and its tap records (once per brick)
    const int col = wave * 2 + (lane >> 5);
    const int vx = kx * kBX + (col & 3), vy = ky * BY + (col >> 2), vz = kz * kBZ + (lane & 31);
    const unsigned vox = (unsigned)(((long long)vx * Y + vy) * Z + vz);              // N < 2^30 (brick_supported)
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int tx[VT], ty[VT];
    unsigned valid = 0;
    {
        const float *Xp = coords + ((long long)b * N + vox) * 3;
        const float c0 = Xp[0], c1 = Xp[1], c2 = Xp[2];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
            w00[v] = t.w00; w01[v] = t.w01; w10[v] = t.w10; w11[v] = t.w11;
            tx[v] = t.rx0; ty[v] = t.ry0;
            if (t.any) valid |= 1u << v;
            const int big = 1 << 30;
            const int xmin = wave_min(t.any ? t.rx0 : big), ymin = wave_min(t.any ? t.ry0 : big);
            const int xmax = wave_max(t.any ? t.rx0 : -big), ymax = wave_max(t.any ? t.ry0 : -big);
            if (lane == 0 && xmax >= xmin) {
                atomicMin(&sh->bbox[v][0], xmin); atomicMin(&sh->bbox[v][1], ymin);
                atomicMax(&sh->bbox[v][2], xmax); atomicMax(&sh->bbox[v][3], ymax);
            }
        }
    }
    __syncthreads();

    // ---- window per view (block-uniform): origin, width, odd row stride, rows, first slot; views packed back to back
    int wx0[VT], wy0[VT], ws[VT], nch[VT + 1], slot0[VT];
    nch[0] = 0;
    int used = 0;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
        const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
        int bw = 0, bh = 0;
        if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }     // taps reach x0+1, y0+1
        // row stride in 16-B slots, forced odd: the 16 lanes of a ds_read_b128 group are 16 consecutive voxels of
        // a column, i.e. ~16 different window rows; an odd stride spreads them over all 16 slots of the bank row
        const int stride = bw | 1;
        const int chunks = (stride * bh + 63) >> 6;                              // 64-slot DMA chunks
        wx0[v] = xmin; wy0[v] = ymin; ws[v] = stride;
        slot0[v] = used;
        used += chunks << 6;
        nch[v + 1] = nch[v] + chunks;
    }
    const bool fits = used <= cap && nch[VT] <= kMaxChunks * NW;
    float *const obase = out + (long long)b * C * N;
    const float4 *const fk = featK + (long long)b * VT * nq * HW;                  // this sample's quad planes

    if (fits) {
        // ---- LDS byte offsets (inside a buffer) of the two tap rows of every view
        int a0[VT], a1[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const bool ok = (valid >> v) & 1u;
            const int s0 = slot0[v] + (ty[v] - wy0[v]) * ws[v] + (tx[v] - wx0[v]);
            a0[v] = ok ? 64 + s0 * 16 : 0;                                       // !ok: weights are 0, read the zero slots
            a1[v] = ok ? a0[v] + ws[v] * 16 : 0;
        }
        // ---- DMA chunks of this wave: chunk c covers 64 consecutive slots of one view's window
        int g_idx[kMaxChunks], l_dst[kMaxChunks];
#pragma unroll
        for (int r = 0; r < kMaxChunks; ++r) {
            const int c = wave + r * NW;
            l_dst[r] = -1;
            g_idx[r] = 0;
            if (c < nch[VT]) {
                int v = 0;
#pragma unroll
                for (int u = 1; u < VT; ++u) v += c >= nch[u] ? 1 : 0;
                int sv = ws[0], ox = wx0[0], oy = wy0[0], c0 = nch[0], s0 = slot0[0];
#pragma unroll
                for (int u = 1; u < VT; ++u) if (v == u) { sv = ws[u]; ox = wx0[u]; oy = wy0[u]; c0 = nch[u]; s0 = slot0[u]; }
                const int j = c - c0, slot = (j << 6) + lane;
                const int py = slot / sv, px = slot - py * sv;
                int gx = ox + px, gy = oy + py;                                  // pad column / rows past the window / outside the
                gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);                     // image: clamp -- those slots only meet zero weights
                gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
                g_idx[r] = (v * nq) * HW + gy * W + gx;
                l_dst[r] = 64 + (s0 + (j << 6)) * 16;
            }
        }
        auto dma = [&](int q) {
            const float4 *src = fk + (long long)q * HW;
            const int boff = (q & 1) * buf_bytes;
#pragma unroll
            for (int r = 0; r < kMaxChunks; ++r)
                if (l_dst[r] >= 0)
                    __builtin_amdgcn_global_load_lds((const void *)(src + g_idx[r]), (lds_void_t *)(smem + uniform(l_dst[r] + boff)), 16, 0, 0);
        };
        auto tap = [&](int addr) -> f32x4 {
            const float4 t = *reinterpret_cast<const float4 *>(smem + addr);
            return f32x4{{t.x, t.y, t.z, t.w}};
        };

        // ---- channel-quad loop: dma(q+1) | sample(q) | wait for dma(q+1) | barrier
        // The barrier orders LDS traffic only; a counted vmcnt leaves this quad's 4 output stores in flight
        // (a __syncthreads() here would drain vmcnt: every wave waiting for its stores 64 times per brick).
        dma(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        for (int q = 0; q < nq; ++q) {
            if (q + 1 < nq) dma(q + 1);
            const int boff = (q & 1) * buf_bytes;
            float s[4][VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const f32x4 a = tap(a0[v] + boff), bb = tap(a0[v] + boff + 16), c = tap(a1[v] + boff), d = tap(a1[v] + boff + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = bilerp(a.v[i], bb.v[i], c.v[i], d.v[i], w00[v], w01[v], w10[v], w11[v]);
            }
            float *oq = obase + (long long)(q * 4) * N;
#pragma unroll
            for (int i = 0; i < 4; ++i) (oq + i * N)[vox] = aggregate<METHOD, VT>(s[i]);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        }
    } else {
        // ---- windows do not fit the LDS pool: sample straight from global memory (clamped taps, zero weights outside)
        int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int x0 = tx[v] < 0 ? 0 : tx[v], y0 = ty[v] < 0 ? 0 : ty[v];
            const int x1 = tx[v] + 1 > W - 1 ? W - 1 : tx[v] + 1, y1 = ty[v] + 1 > H - 1 ? H - 1 : ty[v] + 1;
            const int base = (v * nq) * HW;
            o00[v] = base + y0 * W + x0; o01[v] = base + y0 * W + x1; o10[v] = base + y1 * W + x0; o11[v] = base + y1 * W + x1;
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
            }
            float *oq = obase + (long long)(q * 4) * N;
#pragma unroll
            for (int i = 0; i < 4; ++i) (oq + i * N)[vox] = aggregate<METHOD, VT>(s[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
namespace {
constexpr int kNT = 1024;                             // 1024 voxels per brick: 4 x 8 x 32
constexpr int kBYv = kNT / 128;

int pick_cap()
{
    // two buffers of (64 B zeros + cap 16-B slots) + BrickShared must fit 160 KiB (one block per CU);
    // a block stages at most kMaxChunks * NW chunks of 64 slots per quad
    const int budget = (160 * 1024 - 1024) / 2 - 64;
    int cap = budget / 16;
    const int most = kMaxChunks * (kNT / 64) * 64;
    cap = cap < most ? cap : most;
    return cap & ~63;
}

template <int METHOD, int VT>
hipError_t launch_v(const float4 *featK, const float *proj, const float *coords, float *out, const Problem &p, hipStream_t s)
{
    const int nbx = p.X / kBX, nby = p.Y / kBYv, nbz = p.Z / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int cap = pick_cap();
    const size_t lds = 2 * (64 + (size_t)cap * 16) + sizeof(BrickShared<VT>);
    auto kern = k_fwd_brick<METHOD, VT, kNT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = (total + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kNT), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, cap, total);
    return hipGetLastError();
}

template <int METHOD>
hipError_t launch_m(const float4 *featK, const float *proj, const float *coords, float *out, const Problem &p, hipStream_t s)
{
    switch (p.V) {
    case 2: return launch_v<METHOD, 2>(featK, proj, coords, out, p, s);
    case 4: return launch_v<METHOD, 4>(featK, proj, coords, out, p, s);
    case 8: return launch_v<METHOD, 8>(featK, proj, coords, out, p, s);
    }
    return hipErrorNotSupported;
}
}  // namespace

bool brick_supported(const Problem &p)
{
    if (p.feat_f16 || p.out_f16) return false;                            // fp32 storage only (for now)
    if (p.V != 2 && p.V != 4 && p.V != 8) return false;
    if (p.C % 4 || p.Z % kBZ || p.X % kBX || p.Y % kBYv) return false;
    if ((long long)p.B * p.V * (p.C / 4) * p.H * p.W >= (1ll << 31)) return false;
    if (p.N >= (1ll << 30)) return false;                                 // 32-bit voxel offsets in the stores
    return true;
}

size_t brick_workspace_bytes(const Problem &p)
{
    const size_t n = (size_t)p.B * p.V * p.C * p.H * p.W * sizeof(float);
    return (n + 255) / 256 * 256;
}

hipError_t launch_to_quad_planar(const void *src, void *dst, const Problem &p, hipStream_t s)
{
    if (p.feat_f16 || p.C % 4) return hipErrorNotSupported;
    const int HW = p.H * p.W;
    hipLaunchKernelGGL(k_to_quad_planar, dim3((HW + 255) / 256, p.C / 4, p.B * p.V), dim3(256), 0, s, (const float *)src,
                       (float4 *)dst, p.C, HW);
    return hipGetLastError();
}

hipError_t launch_fwd_brick(const void *featK_, const float *proj, const float *coords, void *out, const Problem &p, hipStream_t s)
{
    if (!brick_supported(p)) return hipErrorNotSupported;
    const float4 *featK = static_cast<const float4 *>(featK_);
    switch (p.method) {
    case AGG_SOFTMAX: return launch_m<AGG_SOFTMAX>(featK, proj, coords, (float *)out, p, s);
    case AGG_SUM: return launch_m<AGG_SUM>(featK, proj, coords, (float *)out, p, s);
    case AGG_MEAN: return launch_m<AGG_MEAN>(featK, proj, coords, (float *)out, p, s);
    case AGG_MAX: return launch_m<AGG_MAX>(featK, proj, coords, (float *)out, p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mvhmr
