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
//   window = per view, the bounding box of the brick's taps (block-wide min/max in LDS), clipped to the
//            LDS budget; staged per channel quad, double-buffered: the next quad's pixels are in flight
//            (global -> registers) while the current quad is sampled, one barrier per quad;
//   taps outside the window (huge perspective spread, exotic cameras) fall back to global loads per lane,
//            so geometry can only cost speed, never correctness;
//   zero padding = out-of-image window pixels are staged as zeros and their tap weights are zero
//            (aggregation.py:55-58, padding_mode='zeros').
// The cross-view aggregate runs in registers exactly as in the gather variant (aggregation.py:71-85).
#include "device_common.h"
#include "kernels.h"

namespace mvhmr {

constexpr int kBZ = 32;            // z extent of a brick: 128-B output runs
constexpr int kBX = 4;
constexpr int kMaxItems = 6;       // window pixels staged per thread per quad

template <int VT>
struct BrickShared {
    int bbox[VT][4];               // xmin, ymin, xmax, ymax of the nw taps (valid voxels only)
    float proj[VT][12];
};

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

template <int METHOD, int VT, int NT, int VPL>
__global__ void __launch_bounds__(NT)
k_fwd_brick(const float4 *__restrict__ featK, const float *__restrict__ proj, const float *__restrict__ coords,
            float *__restrict__ out, int C, int H, int W, int X, int Y, int Z, int nby, int nbz, int bricks_per_sample,
            int pmax, int total_blocks)
{
    constexpr int BY = NT / 128 * VPL;            // columns = NT/32 * VPL, arranged 4 (x) by BY (y)
    extern __shared__ __align__(16) unsigned char smem[];
    // [ 32 B of zeros | buffer 0 : VT * pmax slots | buffer 1 : VT * pmax slots | BrickShared ]
    float4 *slots = reinterpret_cast<float4 *>(smem);
    const int buf_bytes = VT * pmax * 16;
    BrickShared<VT> *sh = reinterpret_cast<BrickShared<VT> *>(smem + 64 + 2 * buf_bytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

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
    if (tid < 4) slots[tid] = make_float4(0.f, 0.f, 0.f, 0.f);          // the always-zero slots (64 B)
    __syncthreads();

    // ---- per-lane voxels and their tap records
    long long vox[VPL];
    float w00[VPL][VT], w01[VPL][VT], w10[VPL][VT], w11[VPL][VT];
    int xy[VPL][VT];                   // raw (x0, y0) of the nw tap, 16 bits each, biased by 1 (x0 >= -1)
    unsigned valid = 0;                // bit (k*VT+v): the sample is not identically zero
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int col = wave * 2 + (lane >> 5) + (NT / 32) * k;
        const int x = kx * kBX + (col & 3), y = ky * BY + (col >> 2), z = kz * kBZ + (lane & 31);
        vox[k] = ((long long)x * Y + y) * Z + z;
        const float *Xp = coords + ((long long)b * N + vox[k]) * 3;
        const float c0 = Xp[0], c1 = Xp[1], c2 = Xp[2];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
            w00[k][v] = t.w00; w01[k][v] = t.w01; w10[k][v] = t.w10; w11[k][v] = t.w11;
            xy[k][v] = ((t.ry0 + 1) << 16) | (t.rx0 + 1);
            if (t.any) {
                valid |= 1u << (k * VT + v);
                atomicMin(&sh->bbox[v][0], t.rx0); atomicMin(&sh->bbox[v][1], t.ry0);
                atomicMax(&sh->bbox[v][2], t.rx0); atomicMax(&sh->bbox[v][3], t.ry0);
            }
        }
    }
    __syncthreads();

    // ---- window per view (block-uniform): origin, width, rows, slot offset
    int wx0[VT], wy0[VT], wp[VT], hp[VT], item0[VT + 1];
    item0[0] = 0;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const int xmin = sh->bbox[v][0], ymin = sh->bbox[v][1], xmax = sh->bbox[v][2], ymax = sh->bbox[v][3];
        int bw = 0, bh = 0;
        if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }     // taps reach x0+1, y0+1
        bw = bw < pmax ? bw : pmax;
        int rows = bw > 0 ? pmax / bw : 0;
        rows = rows < bh ? rows : bh;
        wx0[v] = xmin; wy0[v] = ymin; wp[v] = bw; hp[v] = rows;
        item0[v + 1] = item0[v] + bw * rows;
    }

    // ---- LDS byte address of the nw tap of every (voxel, view); taps outside the window -> global fallback
    int a0[VPL][VT];
    unsigned inwin = 0;
#pragma unroll
    for (int k = 0; k < VPL; ++k)
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int x0 = (xy[k][v] & 0xffff) - 1, y0 = (xy[k][v] >> 16) - 1;
            const int px = x0 - wx0[v], py = y0 - wy0[v];
            const bool ok = (valid >> (k * VT + v)) & 1u;
            const bool in = ok && px + 1 < wp[v] && py + 1 < hp[v];
            a0[k][v] = in ? 64 + (v * pmax + py * wp[v] + px) * 16 : 0;
            if (in || !ok) inwin |= 1u << (k * VT + v);                       // !ok: zero weights, reads the zero slots
        }

    // ---- staging items of this thread: (global float4 index inside one quad plane | -1, LDS slot)
    int g_idx[kMaxItems], l_off[kMaxItems];
    const int n_items = item0[VT];
#pragma unroll
    for (int r = 0; r < kMaxItems; ++r) {
        const int g = tid + r * NT;
        g_idx[r] = -2;                                                        // -2: no item
        l_off[r] = 0;
        if (g < n_items) {
            int v = 0;
#pragma unroll
            for (int u = 1; u < VT; ++u) v += g >= item0[u] ? 1 : 0;
            int wv = wp[0], ox = wx0[0], oy = wy0[0], i0 = item0[0];
#pragma unroll
            for (int u = 1; u < VT; ++u) if (v == u) { wv = wp[u]; ox = wx0[u]; oy = wy0[u]; i0 = item0[u]; }
            const int i = g - i0, py = i / wv, px = i - py * wv;
            const int gx = ox + px, gy = oy + py;
            const bool img = gx >= 0 && gx < W && gy >= 0 && gy < H;
            g_idx[r] = img ? (v * nq) * HW + gy * W + gx : -1;                // -1: outside the image -> zeros
            l_off[r] = 64 + (v * pmax + i) * 16;
        }
    }
    const float4 *fk = featK + (long long)b * VT * nq * HW;                  // this sample's quad planes

    auto fetch = [&](int q, float4 (&pre)[kMaxItems]) {
#pragma unroll
        for (int r = 0; r < kMaxItems; ++r) {
            pre[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_idx[r] >= 0) pre[r] = fk[g_idx[r] + q * HW];
        }
    };
    auto stage = [&](int bufsel, const float4 (&pre)[kMaxItems]) {
#pragma unroll
        for (int r = 0; r < kMaxItems; ++r)
            if (g_idx[r] != -2) *reinterpret_cast<float4 *>(smem + l_off[r] + bufsel * buf_bytes) = pre[r];
    };
    auto tap = [&](int addr) -> f32x4 {
        const float4 t = *reinterpret_cast<const float4 *>(smem + addr);
        return f32x4{{t.x, t.y, t.z, t.w}};
    };
    auto gtap = [&](int q, int v, int x, int y) -> f32x4 {                    // clamped: zero-weight taps may sit outside
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
        const float4 t = fk[(v * nq + q) * HW + y * W + x];
        return f32x4{{t.x, t.y, t.z, t.w}};
    };

    auto compute = [&](int q, int bufsel) {
        const int boff = bufsel * buf_bytes;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            float s[4][VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                f32x4 a, bb, c, d;
                if ((inwin >> (k * VT + v)) & 1u) {
                    const int base = a0[k][v] ? a0[k][v] + boff : 0;          // 0 = the zero slots (not double-buffered)
                    const int row1 = a0[k][v] ? base + wp[v] * 16 : 0;
                    a = tap(base); bb = tap(base + 16); c = tap(row1); d = tap(row1 + 16);
                } else {
                    const int x0 = (xy[k][v] & 0xffff) - 1, y0 = (xy[k][v] >> 16) - 1;
                    a = gtap(q, v, x0, y0); bb = gtap(q, v, x0 + 1, y0); c = gtap(q, v, x0, y0 + 1); d = gtap(q, v, x0 + 1, y0 + 1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = bilerp(a.v[i], bb.v[i], c.v[i], d.v[i], w00[k][v], w01[k][v], w10[k][v], w11[k][v]);
            }
            float *o = out + ((long long)b * C + q * 4) * N + vox[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i * N] = aggregate<METHOD, VT>(s[i]);
        }
    };

    // ---- channel-quad loop, double-buffered: fetch(q+1) | compute(q) | stage(q+1) | barrier
    float4 pre[kMaxItems];
    fetch(0, pre);
    stage(0, pre);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const bool more = q + 1 < nq;
        if (more) fetch(q + 1, pre);
        compute(q, q & 1);
        if (more) stage((q + 1) & 1, pre);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ host side
namespace {
constexpr int kNT = 1024, kVPL = 1;                   // 1024 voxels per brick: 4 x 8 x 32
constexpr int kBYv = kNT / 128 * kVPL;

int pick_pmax(int V)
{
    // two buffers of V * pmax 16-B slots + 64 B of zeros + BrickShared must fit 160 KiB (one block per CU)
    const int budget = 160 * 1024 - 64 - 1024;
    int pmax = budget / (2 * V * 16);
    const int cap = (kMaxItems * kNT) / V;               // what the block can stage per quad
    pmax = pmax < cap ? pmax : cap;
    return pmax & ~1;
}

template <int METHOD, int VT>
hipError_t launch_v(const float4 *featK, const float *proj, const float *coords, float *out, const Problem &p, hipStream_t s)
{
    const int nbx = p.X / kBX, nby = p.Y / kBYv, nbz = p.Z / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int pmax = pick_pmax(VT);
    const size_t lds = 64 + 2 * (size_t)VT * pmax * 16 + sizeof(BrickShared<VT>);
    auto kern = k_fwd_brick<METHOD, VT, kNT, kVPL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = (total + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kNT), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, pmax, total);
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
    if (p.N * 4 >= (1ll << 40)) return false;
    if (p.H > 32000 || p.W > 32000) return false;                         // 16-bit packed tap coordinates
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
