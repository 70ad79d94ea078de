// Host side of the brick forward (kernel: brick_fwd_kernel.h, instantiated per aggregation method in unproject_brick_fwd_m*.hip),
// the layout passes that build the staged fp32 copies, and the geometry gate.
#include "brick_fwd_ws.h"

namespace mvhmr {

// ------------------------------------------------------------------------------------------------- layout passes
// features (BV, C, H, W) fp32 / fp16 -> fp32 (BV, C/4, W, H, 4): COLUMN-major quad-planar, what the brick forward stages
// (MVHMR_LAYOUT_QUAD).  One block = 4 channels x (32 x 32) pixels turned through LDS: reads are 128-B runs along x, writes
// 512-B runs along y.  fp16 features are widened here once instead of per tap in the kernel.
template <typename TF>
__global__ void __launch_bounds__(256)
k_to_quad_planar_t(const TF *__restrict__ src, float4 *__restrict__ dst, int C, int H, int W, float scale, Gate gate)
{
    if (gated_off(gate)) return;
    __shared__ float tile[4][32][33];                                           // [c][y][x], x padded: conflict-free both ways
    const long long bv = blockIdx.z;
    const int q = blockIdx.y;
    const int tiles_x = (W + 31) >> 5;
    const int x0 = (blockIdx.x % tiles_x) << 5, y0 = (blockIdx.x / tiles_x) << 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                     // 8 rows per pass
    const TF *s = src + (bv * C + q * 4) * (long long)H * W;
    const int nc = C - q * 4 < 4 ? C - q * 4 : 4;                                // the last quad of C % 4 != 0 channels: the missing ones are zeros
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + ty + 8 * i, x = x0 + tx;
            if (y < H && x < W) tile[c][ty + 8 * i][tx] = c < nc ? to_f32<TF>(s[(long long)c * H * W + (long long)y * W + x]) * scale : 0.f;
        }
    __syncthreads();
    float4 *d = dst + (bv * ((C + 3) >> 2) + q) * (long long)H * W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int xx = ty + 8 * i, y = y0 + tx, x = x0 + xx;                    // lanes run along y
        if (y < H && x < W) d[(long long)x * H + y] = make_float4(tile[0][tx][xx], tile[1][tx][xx], tile[2][tx][xx], tile[3][tx][xx]);
    }
}

// Same pass for maps whose 32-row bands fit LDS: a band of one channel quad is 4 x 32 x W contiguous floats per channel, read
// linearly (1-KiB runs per wave instruction instead of 128-B tile rows) and written as 512-B runs along y.
template <typename TF>
__global__ void __launch_bounds__(512)
k_to_quad_planar_t_band(const TF *__restrict__ src, float4 *__restrict__ dst, int C, int H, int W, int src_aligned, float scale, Gate gate)
{
    if (gated_off(gate)) return;
    extern __shared__ float band[];                                              // [c][y][x], row stride W | 1
    const int ldw = W | 1;
    const long long bv = blockIdx.z;
    const int q = blockIdx.y, y0 = blockIdx.x << 5;
    const int rows = H - y0 < 32 ? H - y0 : 32;
    const long long plane = (long long)H * W;
    const TF *s = src + (bv * C + q * 4) * plane + (long long)y0 * W;
    const int n = rows * W;
    const int nc = C - q * 4 < 4 ? C - q * 4 : 4;                                // the last quad of C % 4 != 0 channels: the missing ones are zeros
    for (int c = nc; c < 4; ++c)
        for (int i = threadIdx.x; i < n; i += 512) band[(c * 32 + i / W) * ldw + i % W] = 0.f;
    // vector loads only from an aligned base (src_aligned: 16 B for fp32, 8 B for fp16 -- a slice of a flat buffer or a raw C-ABI
    // pointer may start anywhere: ADVICE r03)
    if (sizeof(TF) == 4 && (W & 3) == 0 && (plane & 3) == 0 && src_aligned) {
        // 16-B loads: 4 consecutive x of one row (W % 4 == 0 keeps them inside a row and aligned)
        for (int c = 0; c < nc; ++c)
            for (int i = threadIdx.x * 4; i < n; i += 512 * 4) {
                const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(s) + (long long)c * plane + i);
                const int y = i / W, x = i - y * W;
                float *b = band + (c * 32 + y) * ldw + x;
                b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
            }
    } else if (sizeof(TF) == 2 && (W & 3) == 0 && (plane & 3) == 0 && src_aligned) {
        // fp16 features: 8-B loads of 4 consecutive x (2-B loads ran this pass at 3.8 TB/s instead of the fp32 form's 5.9)
        for (int c = 0; c < nc; ++c)
            for (int i = threadIdx.x * 4; i < n; i += 512 * 4) {
                const f32x4 t = Vec4<__half>::load(reinterpret_cast<const __half *>(s) + (long long)c * plane + i);
                const int y = i / W, x = i - y * W;
                float *b = band + (c * 32 + y) * ldw + x;
                b[0] = t.v[0]; b[1] = t.v[1]; b[2] = t.v[2]; b[3] = t.v[3];
            }
    } else {
        for (int c = 0; c < nc; ++c)
            for (int i = threadIdx.x; i < n; i += 512) {
                const int y = i / W, x = i - y * W;
                band[(c * 32 + y) * ldw + x] = to_f32<TF>(s[(long long)c * plane + i]);
            }
    }
    __syncthreads();
    float4 *d = dst + (bv * ((C + 3) >> 2) + q) * plane;
    const int ty = threadIdx.x & 31;
    if (ty < rows)
        for (int x = threadIdx.x >> 5; x < W; x += 16)
            d[(long long)x * H + y0 + ty] = make_float4(band[ty * ldw + x] * scale, band[(32 + ty) * ldw + x] * scale, band[(64 + ty) * ldw + x] * scale,
                                                        band[(96 + ty) * ldw + x] * scale);   // scale = 1 (exact) or log2(e) for the softmax forward
}

hipError_t launch_to_quad_planar_t(const void *src, void *dst, const Problem &p, hipStream_t s, bool brick_side)
{
    const float scale = p.feat_log2e ? kLog2e : 1.f;
    const Gate gate = make_gate(p, brick_side);
    const size_t band_bytes = (size_t)4 * 32 * (p.W | 1) * sizeof(float);
    if (band_bytes <= 64 * 1024) {                                               // 2+ blocks per CU
        const dim3 grid((p.H + 31) / 32, p.C4 / 4, p.B * p.V);
        const int aligned = (reinterpret_cast<uintptr_t>(src) % (p.feat_f16 ? 8 : 16)) == 0;
        if (p.feat_f16) hipLaunchKernelGGL(k_to_quad_planar_t_band<__half>, grid, dim3(512), band_bytes, s, (const __half *)src, (float4 *)dst, p.C, p.H, p.W, aligned, scale, gate);
        else hipLaunchKernelGGL(k_to_quad_planar_t_band<float>, grid, dim3(512), band_bytes, s, (const float *)src, (float4 *)dst, p.C, p.H, p.W, aligned, scale, gate);
        return hipGetLastError();
    }
    const dim3 grid(((p.W + 31) / 32) * ((p.H + 31) / 32), p.C4 / 4, p.B * p.V);
    if (p.feat_f16) hipLaunchKernelGGL(k_to_quad_planar_t<__half>, grid, dim3(256), 0, s, (const __half *)src, (float4 *)dst, p.C, p.H, p.W, scale, gate);
    else hipLaunchKernelGGL(k_to_quad_planar_t<float>, grid, dim3(256), 0, s, (const float *)src, (float4 *)dst, p.C, p.H, p.W, scale, gate);
    return hipGetLastError();
}

// Channels-last features (BV, H, W, C) -> the same column-major quad-planar copy.  One block = 16 image rows of one column, all
// quads 64 at a time: read as 1-KiB runs along the channels (64 quads x 16 B), turned through LDS, written as 256-B runs along y.
template <typename TF>
__global__ void __launch_bounds__(256)
k_channels_last_to_quad_t(const TF *__restrict__ src, float4 *__restrict__ dst, int C, int H, int W, Gate gate)
{
    if (gated_off(gate)) return;
    __shared__ float4 tile[64][17];                                              // [quad][row], padded: the column-wise reads hit distinct banks
    const long long bv = blockIdx.z;
    const int w = blockIdx.y, h0 = blockIdx.x << 4, nq = C >> 2;
    for (int qc = 0; qc < nq; qc += 64) {
        for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
            const int hh = idx >> 6, ql = idx & 63;
            if (h0 + hh < H && qc + ql < nq) {
                const TF *s = src + ((bv * H + h0 + hh) * W + w) * (long long)C + 4 * (qc + ql);
                tile[ql][hh] = make_float4(to_f32<TF>(s[0]), to_f32<TF>(s[1]), to_f32<TF>(s[2]), to_f32<TF>(s[3]));
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
            const int ql = idx >> 4, hh = idx & 15;
            if (h0 + hh < H && qc + ql < nq) dst[((bv * nq + qc + ql) * W + w) * (long long)H + h0 + hh] = tile[ql][hh];
        }
        __syncthreads();
    }
}

hipError_t launch_channels_last_to_quad_planar_t(const void *src, void *dst, const Problem &p, hipStream_t s)
{
    if (p.C % 4 || p.W > 65535 || p.B * p.V > 65535) return hipErrorNotSupported;
    const dim3 grid((p.H + 15) / 16, p.W, p.B * p.V);
    const Gate gate = make_gate(p, true);
    if (p.feat_f16) hipLaunchKernelGGL(k_channels_last_to_quad_t<__half>, grid, dim3(256), 0, s, (const __half *)src, (float4 *)dst, p.C, p.H, p.W, gate);
    else hipLaunchKernelGGL(k_channels_last_to_quad_t<float>, grid, dim3(256), 0, s, (const float *)src, (float4 *)dst, p.C, p.H, p.W, gate);
    return hipGetLastError();
}

size_t brick_workspace_bytes(const Problem &p)
{
    const size_t n = (size_t)p.B * p.V * p.C4 * p.H * p.W * sizeof(float);            // (C + 3) / 4 quads per view: the last one zero-padded
    return (n + 255) / 256 * 256;
}

// ------------------------------------------------------------------------------------------------- brick geometry
int fwd_lds_slots() { return (160 * 1024 - 1024) / 16; }
static int fwd_threads(int V) { return V > 4 ? 512 : 1024; }           // 8 view slots: 256 VGPRs per lane
// 8 x 8 x 32 bricks (two voxels per lane) when x allows; 8 views keep 4 x 4 x 32: their eight windows of a doubled brick (mean
// ~4 300 slots, max ~6 000 at the configs[3] geometry) overflow the 2-deep ring (4 928) for most bricks
// (volumes need not divide into bricks: lanes outside the volume idle.  Two voxels per lane whenever the volume is wider than one brick)
int brick_fwd_nvox(const Problem &p) { return (p.V <= 4 && p.X > kBX) ? 2 : 1; }
// 8 views: 1024-thread blocks on 4 x 8 x 32 bricks with the views staged in two groups of four (brick_fwd_groups.h) when y divides
#ifndef MVHMR_NO_GROUPS
#define MVHMR_NO_GROUPS 0
#endif
bool brick_fwd_grouped(const Problem &p) { return !MVHMR_NO_GROUPS && p.V > 4; }     // 5 ... 8 views (the missing ones absent)

bool brick_fwd_supported(const Problem &p)
{
    if (p.V < 1 || p.V > 8) return false;                                 // 1 / 3 / 5 / 6 / 7 views: the next larger kernel, missing views absent
    if (p.C < 4 || ((p.C & 3) && p.B > 65535)) return false;              // r05: C % 4 != 0 -- the whole quads through the fast loops, the rest per voxel (k_fwd_tail: grid.y = B)
    // r04: any X, Y, Z -- bricks that stick out of the volume idle their outside lanes.  16-bit volumes store z pairs: Z even.
    if ((p.out_f16 || p.out_bf16) && (p.Z & 1)) return false;
    if ((long long)p.B * p.V * (p.C4 / 4) * p.H * p.W >= (1ll << 31)) return false;
    if (p.N >= (1ll << 28)) return false;                                 // 32-bit byte offsets inside one quad of the output
    return true;
}

// the wave-specialised kernel (brick_fwd_ws.h) serves 3 / 4 views with an fp32 volume when the launch fills the chip
#ifndef MVHMR_NO_WS
#define MVHMR_NO_WS 0
#endif
bool brick_fwd_ws_shape(const Problem &p) { return !MVHMR_NO_WS && brick_fwd_supported(p) && brick_fwd_ws_shape_impl(p); }
// its softmax reads a copy of the features multiplied by log2(e): the layout pass in front of it scales (Problem::feat_log2e)
bool brick_fwd_prescales(const Problem &p) { return brick_fwd_ws_shape(p) && p.method == AGG_SOFTMAX; }

GateGeom brick_fwd_gate_geom(const Problem &p)
{
    if (brick_fwd_ws_shape(p)) {
        GateGeom g;
        g.bx = 8; g.by = 8; g.bz = kBZ; g.column_major = 1; g.parity_rows = 1; g.view_group = 0;
        g.cap_slots = kWsCapSlots;
        g.max_chunks = kWsChunkTotal;
        return g;
    }
    const int nt = brick_fwd_grouped(p) ? 1024 : fwd_threads(p.V);
    GateGeom g;
    g.bx = kBX * brick_fwd_nvox(p); g.by = nt / 128; g.bz = kBZ; g.column_major = 1;
    g.parity_rows = kFwdLay == 1 ? 1 : 0;
    g.view_group = brick_fwd_grouped(p) ? 4 : 0;
    g.cap_slots = fwd_cap2(fwd_lds_slots());                              // the 2-deep ring still stages through LDS
    g.max_chunks = brick_fwd_grouped(p) ? 4 * (nt / 64) : brick_chunks_per_wave(nt) * (nt / 64);
    return g;
}

int brick_count(const Problem &p, const GateGeom &g)
{
    return ((p.X + g.bx - 1) / g.bx) * ((p.Y + g.by - 1) / g.by) * ((p.Z + g.bz - 1) / g.bz) * p.B;
}

// ---- geometry gate: one thread per brick projects the brick's 8 corner voxels into every view and sizes the pooled windows the
// brick kernels would need (same arithmetic as their prologue: bbox + 2, odd line stride, 64-slot chunks).  Voxel centres are
// affine in the index for every volume the caller builds, so the corners bound the brick's taps; only speed depends on it.
__global__ void __launch_bounds__(256)
k_brick_gate(const float *__restrict__ proj, const Coords coords, int *__restrict__ count, int V, int H, int W, int X,
             int Y, int Z, GateGeom g, int nbx, int nby, int nbz, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int bps = nbx * nby * nbz;
    const int b = i / bps, r = i % bps;
    const int kz = r % nbz, ky = (r / nbz) % nby, kx = r / (nbz * nby);
    const long long N = (long long)X * Y * Z;
    int used = 0, chunks_all = 0, max_stride = 0, used_max = 0, chunks_max = 0;
    for (int v = 0; v < V; ++v) {
        if (g.view_group && v % g.view_group == 0) { used = 0; chunks_all = 0; }    // a new group of views starts from an empty buffer
        const float *P = proj + ((long long)b * V + v) * 12;
        float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
        bool front = true;
        for (int c = 0; c < 8; ++c) {
            int vx = kx * g.bx + ((c & 1) ? g.bx - 1 : 0), vy = ky * g.by + ((c & 2) ? g.by - 1 : 0), vz = kz * g.bz + ((c & 4) ? g.bz - 1 : 0);
            vx = vx < X ? vx : X - 1; vy = vy < Y ? vy : Y - 1; vz = vz < Z ? vz : Z - 1;      // bricks that stick out of the volume
            float Xp[3];
            voxel_xyz(coords, b, N, ((long long)vx * Y + vy) * Z + vz, Xp[0], Xp[1], Xp[2]);
            const float a = P[0] * Xp[0] + P[1] * Xp[1] + P[2] * Xp[2] + P[3];
            const float bb = P[4] * Xp[0] + P[5] * Xp[1] + P[6] * Xp[2] + P[7];
            const float z = P[8] * Xp[0] + P[9] * Xp[1] + P[10] * Xp[2] + P[11];
            if (!(z > 0.f)) { front = false; continue; }
            const float ix = (a / z) / (float)H * (float)(W - 1), iy = (bb / z) / (float)W * (float)(H - 1);   // quirk Q1 as in make_taps
            xmin = fminf(xmin, ix); xmax = fmaxf(xmax, ix); ymin = fminf(ymin, iy); ymax = fmaxf(ymax, iy);
        }
        if (!front || xmax < xmin) continue;                                     // behind a camera: the kernels decide per block
        const float x0 = fmaxf(floorf(xmin), -1.f), x1 = fminf(floorf(xmax), (float)(W - 1));
        const float y0 = fmaxf(floorf(ymin), -1.f), y1 = fminf(floorf(ymax), (float)(H - 1));
        if (x1 < x0 || y1 < y0) continue;                                        // wholly outside the image
        const int bw = (int)(x1 - x0) + 2, bh = (int)(y1 - y0) + 2;
        int stride = (g.column_major ? bh : bw) | 1;
        const int lines = g.column_major ? bw : bh;
        if (g.parity_rows) {                                                     // the forward's parity-split columns: origin row even, 2 hp slots
            const int y0e = (int)y0 & ~1;
            stride = 2 * (((int)y1 + 3 - y0e) >> 1);
        }
        const int chunks = (stride * lines + 63) >> 6;
        used += chunks << 6;
        chunks_all += chunks;
        max_stride = stride > max_stride ? stride : max_stride;
        used_max = used > used_max ? used : used_max;
        chunks_max = chunks_all > chunks_max ? chunks_all : chunks_max;
    }
    const bool fits = used_max <= g.cap_slots && chunks_max <= g.max_chunks && max_stride + 2 <= kZeroSlots;
    if (!fits) atomicAdd(count, 1);
}

hipError_t launch_brick_gate(const float *proj, const Coords &coords, int *count, const GateGeom &g, const Problem &p, hipStream_t s)
{
    const int nbx = (p.X + g.bx - 1) / g.bx, nby = (p.Y + g.by - 1) / g.by, nbz = (p.Z + g.bz - 1) / g.bz, total = nbx * nby * nbz * p.B;
    hipLaunchKernelGGL(k_brick_gate, dim3((total + 255) / 256), dim3(256), 0, s, proj, coords, count, p.V, p.H, p.W, p.X, p.Y, p.Z, g,
                       nbx, nby, nbz, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------- dispatch
extern template hipError_t launch_fwd_method<AGG_SOFTMAX>(const void *, const float *, const Coords &, void *, const Problem &, int, hipStream_t);
extern template hipError_t launch_fwd_method<AGG_SUM>(const void *, const float *, const Coords &, void *, const Problem &, int, hipStream_t);
extern template hipError_t launch_fwd_method<AGG_MEAN>(const void *, const float *, const Coords &, void *, const Problem &, int, hipStream_t);
extern template hipError_t launch_fwd_method<AGG_MAX>(const void *, const float *, const Coords &, void *, const Problem &, int, hipStream_t);

// ---- C % 4 != 0: the brick kernels' loops run the C / 4 whole channel quads; the last, partial quad of the staged copy (its missing
// channels are zeros) is sampled per voxel from global memory by this kernel, launched behind them -- one thread per voxel, the 1 ... 3
// channels of the quad, the gather kernels' speed for 1 of (C + 3) / 4 quads.  (Inside the brick kernels the same code as a cold tail
// cost their hot loops registers: spills in k_fwd_brick's quad loop, which the build gate refuses.)
template <int METHOD, int VT, typename TO>
__global__ void __launch_bounds__(256)
k_fwd_tail(const float4 *__restrict__ featK, const float *__restrict__ proj, const Coords coords, TO *__restrict__ out, int C, int H, int W, long long N,
           int nv, float unscale, Gate gate)
{
    if (gated_off(gate)) return;
    __shared__ float sproj[VT][12];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < VT * 12) sproj[tid / 12][tid % 12] = tid < nv * 12 ? proj[((long long)b * nv) * 12 + tid] : 0.f;
    __syncthreads();
    const long long n = (long long)blockIdx.x * 256 + tid;
    if (n >= N) return;
    const int nqv = (C + 3) >> 2;
    fwd_global_voxel<METHOD, VT, TO>(featK + (long long)b * nv * nqv * ((long long)H * W), out + (long long)b * C * N, sproj, coords, b, N, (unsigned)n, C >> 2, nqv,
                                     nqv, C, H, W, nv, unscale);
}

template <int METHOD, typename TO>
static hipError_t launch_fwd_tail_views(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s)
{
    const dim3 grid((unsigned)((p.N + 255) / 256), (unsigned)p.B);
    const float unscale = p.feat_log2e ? kLn2 : 1.f;
    const Gate gate = make_gate(p, true);
    switch (brick_view_slots(p.V)) {
    case 2: hipLaunchKernelGGL((k_fwd_tail<METHOD, 2, TO>), grid, dim3(256), 0, s, featK, proj, coords, out, p.C, p.H, p.W, p.N, p.V, unscale, gate); break;
    case 4: hipLaunchKernelGGL((k_fwd_tail<METHOD, 4, TO>), grid, dim3(256), 0, s, featK, proj, coords, out, p.C, p.H, p.W, p.N, p.V, unscale, gate); break;
    case 8: hipLaunchKernelGGL((k_fwd_tail<METHOD, 8, TO>), grid, dim3(256), 0, s, featK, proj, coords, out, p.C, p.H, p.W, p.N, p.V, unscale, gate); break;
    default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

template <int METHOD>
static hipError_t launch_fwd_tail(const void *featK, const float *proj, const Coords &coords, void *out, const Problem &p, hipStream_t s)
{
    const float4 *fk = static_cast<const float4 *>(featK);
    return p.out_f16    ? launch_fwd_tail_views<METHOD, __half>(fk, proj, coords, (__half *)out, p, s)
           : p.out_bf16 ? launch_fwd_tail_views<METHOD, bf16_t>(fk, proj, coords, (bf16_t *)out, p, s)
                        : launch_fwd_tail_views<METHOD, float>(fk, proj, coords, (float *)out, p, s);
}

// featK: column-major quad-planar fp32 copy of the features (launch_to_quad_planar_t)
hipError_t launch_fwd_brick(const void *featK, const float *proj, const Coords &coords, void *out, const Problem &p, hipStream_t s)
{
    if (!brick_fwd_supported(p)) return hipErrorNotSupported;
    const int nvox = brick_fwd_nvox(p);
    hipError_t e = hipErrorInvalidValue;
    switch (p.method) {
    case AGG_SOFTMAX: e = launch_fwd_method<AGG_SOFTMAX>(featK, proj, coords, out, p, nvox, s); break;
    case AGG_SUM: e = launch_fwd_method<AGG_SUM>(featK, proj, coords, out, p, nvox, s); break;
    case AGG_MEAN: e = launch_fwd_method<AGG_MEAN>(featK, proj, coords, out, p, nvox, s); break;
    case AGG_MAX: e = launch_fwd_method<AGG_MAX>(featK, proj, coords, out, p, nvox, s); break;
    }
    if (e != hipSuccess || !(p.C & 3)) return e;
    switch (p.method) {                                                          // the last, partial quad
    case AGG_SOFTMAX: return launch_fwd_tail<AGG_SOFTMAX>(featK, proj, coords, out, p, s);
    case AGG_SUM: return launch_fwd_tail<AGG_SUM>(featK, proj, coords, out, p, s);
    case AGG_MEAN: return launch_fwd_tail<AGG_MEAN>(featK, proj, coords, out, p, s);
    case AGG_MAX: return launch_fwd_tail<AGG_MAX>(featK, proj, coords, out, p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace mvhmr
