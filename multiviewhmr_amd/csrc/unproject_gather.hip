// "gather" variant of the fused un-projection: the shape-agnostic kernel pair.
//
// Mapping (CDNA4, wave64):
//   block   = 32 consecutive voxels x one group of up to 256 channels of one sample (256 threads, 4 waves)
//   phase 1 = one thread per (voxel, view): project, build the 4 bilinear taps -> 32-byte record in LDS
//             (done once, reused by every channel: models/aggregation.py:38-54 hoisted out of the channel dim)
//   phase 2 = one wave per voxel, one LANE per 4 consecutive channels: the tap records are wave-uniform
//             (SGPRs), each tap is ONE coalesced 1-KiB read of a channels-last feature row, the
//             cross-view aggregate runs in registers (no (V,C,N) intermediate, aggregation.py:32,68,78-83)
//   phase 3 = the (voxel x channel) tile is turned through LDS so every store writes 128-B runs of
//             consecutive voxels of one channel of the (B,C,X,Y,Z) output
// Feature reads come from L2 / Infinity Cache (each (b,v) map is re-read by many blocks), the output is
// written exactly once.  The brick variant (brick_fwd_kernel.h, unproject_brick_bwd.hip) replaces the L2 gather by LDS patches.
#include "device_common.h"
#include "kernels.h"

namespace mvhmr {

constexpr int kTileVox = 32;          // voxels per block: 128-B output runs; keeps the LDS tile at 33 KB (3 blocks per CU)
constexpr int kGroupQuads = 64;      // 64 lanes x 4 channels
constexpr int kGroupCh = 256;
constexpr unsigned kGatedTiles = 256; // grid.x of a geometry-gated launch (each block then loops over tiles)

struct alignas(16) TapRec {
    int o00, o01, o10, o11;          // element offsets (pixel * C4) inside one (b,v) channels-last map
    float w00, w01, w10, w11;
};

__device__ __forceinline__ void build_records(TapRec *recs, const float *__restrict__ proj,
                                              const Coords &coords, int b, int V, long long n0,
                                              long long N, int H, int W, int C4)
{
    for (int idx = threadIdx.x; idx < kTileVox * V; idx += blockDim.x) {
        const int v = idx / kTileVox, j = idx % kTileVox;
        long long n = n0 + j;
        n = n < N ? n : N - 1;       // tail voxels are computed and dropped
        float X0, X1, X2;
        voxel_xyz(coords, b, N, n, X0, X1, X2);
        const Taps t = make_taps(proj + ((long long)b * V + v) * 12, X0, X1, X2, H, W);
        TapRec r;
        r.o00 = (t.y0 * W + t.x0) * C4;
        r.o01 = (t.y0 * W + t.x1) * C4;
        r.o10 = (t.y1 * W + t.x0) * C4;
        r.o11 = (t.y1 * W + t.x1) * C4;
        r.w00 = t.w00; r.w01 = t.w01; r.w10 = t.w10; r.w11 = t.w11;
        recs[j * V + v] = r;
    }
}

// output stores bypass-ish the caches (nt): the volume is written once and must not displace the feature rows in L2
__device__ __forceinline__ void store_streaming(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_streaming(__half *p, float v)
{
    __builtin_nontemporal_store(__half_as_ushort(from_f32<__half>(v)), reinterpret_cast<unsigned short *>(p));   // fp32 first, then fp16
}

__device__ __forceinline__ void store_streaming(bf16_t *p, float v)
{
    __builtin_nontemporal_store(__builtin_bit_cast(unsigned short, (bf16_t)v), reinterpret_cast<unsigned short *>(p));
}

struct UTap { int o00, o01, o10, o11; float w00, w01, w10, w11; };
__device__ __forceinline__ UTap uniform_rec(const TapRec &r)
{
    UTap u;
    u.o00 = uniform(r.o00); u.o01 = uniform(r.o01); u.o10 = uniform(r.o10); u.o11 = uniform(r.o11);
    u.w00 = uniform(r.w00); u.w01 = uniform(r.w01); u.w10 = uniform(r.w10); u.w11 = uniform(r.w11);
    return u;
}

// ------------------------------------------------------------------------------------------ forward
template <typename TF, typename TO, int METHOD, int VT>
__global__ void __launch_bounds__(256)
k_fwd_gather(const TF *__restrict__ featT, const float *__restrict__ proj, const Coords coords,
             TO *__restrict__ out, int Vrt, int C, int C4, int H, int W, long long N, int tstride, Gate gate)
{
    if (gated_off(gate)) return;
    const int V = VT > 0 ? VT : Vrt;
    extern __shared__ __align__(16) unsigned char smem[];
    TapRec *recs = reinterpret_cast<TapRec *>(smem);
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem + sizeof(TapRec) * kTileVox * V);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, cg = blockIdx.z;
    // one tile per iteration; the grid covers every tile (one iteration) unless the launch is geometry-gated, where a
    // short grid keeps the cost of a gated-off launch at a few thousand empty blocks
    for (long long n0 = (long long)blockIdx.x * kTileVox; n0 < N; n0 += (long long)gridDim.x * kTileVox) {
    const long long mapsz = (long long)H * W * C4;
    const int Q = C4 >> 2;

    build_records(recs, proj, coords, b, V, n0, N, H, W, C4);
    __syncthreads();

    int q = cg * kGroupQuads + lane;
    const bool q_active = q < Q;
    q = q_active ? q : Q - 1;                                // idle lanes shadow the last quad and write nothing
    const TF *fb = featT + (long long)b * V * mapsz + q * 4;

    for (int jj = 0; jj < kTileVox / 4; ++jj) {
        const int j = wave * (kTileVox / 4) + jj;
        f32x4 o;
        if constexpr (VT > 0) {
            float s[4][VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const UTap u = uniform_rec(recs[j * VT + v]);
                const TF *fv = fb + v * mapsz;
                // a sample that is identically zero (z <= 0, or all four taps outside the map) reads nothing: its dummy taps
                // must not turn a non-finite pixel (0, 0) into 0 * Inf (wave-uniform branch)
                if (u.w00 == 0.f && u.w01 == 0.f && u.w10 == 0.f && u.w11 == 0.f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[i][v] = 0.f;
                    continue;
                }
                const f32x4 a = Vec4<TF>::load(fv + u.o00), bb = Vec4<TF>::load(fv + u.o01);
                const f32x4 c = Vec4<TF>::load(fv + u.o10), d = Vec4<TF>::load(fv + u.o11);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = bilerp(a.v[i], bb.v[i], c.v[i], d.v[i], u.w00, u.w01, u.w10, u.w11);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) o.v[i] = aggregate<METHOD, VT>(s[i]);
        } else {
            RunningAgg<METHOD> ra[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i].init();
            for (int v = 0; v < V; ++v) {
                const UTap u = uniform_rec(recs[j * V + v]);
                const TF *fv = fb + v * mapsz;
                if (u.w00 == 0.f && u.w01 == 0.f && u.w10 == 0.f && u.w11 == 0.f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ra[i].push(0.f);
                    continue;
                }
                const f32x4 a = Vec4<TF>::load(fv + u.o00), bb = Vec4<TF>::load(fv + u.o01);
                const f32x4 c = Vec4<TF>::load(fv + u.o10), d = Vec4<TF>::load(fv + u.o11);
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i].push(bilerp(a.v[i], bb.v[i], c.v[i], d.v[i], u.w00, u.w01, u.w10, u.w11));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) o.v[i] = ra[i].result(V);
        }
        if (q_active) tile[j * tstride + lane] = o;
    }
    __syncthreads();

    // lanes 0..31 = voxels of the tile, the two half-waves take alternate channel quads: each store instruction writes
    // two 128-B runs (32 consecutive voxels of two channels)
    const int vl = lane & (kTileVox - 1), half = lane / kTileVox;
    const long long n = n0 + vl;
    if (n < N) {
        for (int qq = wave * 16 + half; qq < wave * 16 + 16; qq += 64 / kTileVox) {
            const int cq = cg * kGroupQuads + qq;
            if (cq >= Q) break;
            const f32x4 t = tile[vl * tstride + qq];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = cq * 4 + i;
                if (c < C) store_streaming(&out[((long long)b * C + c) * N + n], t.v[i]);   // keep L2 for the features
            }
        }
    }
    __syncthreads();                                          // records and tile are rebuilt by the next iteration
    }
}

// ------------------------------------------------------------------------------------------ backward
// lane <-> channels {l, l+64, l+128, l+192} of the group so that every float-atomic wave instruction
// adds 256 contiguous bytes of the channels-last gradient row (the full-rate shape on gfx950).
template <typename TF, typename TO, int METHOD, int VT>
__global__ void __launch_bounds__(256)
k_bwd_gather(const TO *__restrict__ grad_out, const TF *__restrict__ featT, const float *__restrict__ proj,
             const Coords coords, float *__restrict__ gradT, int Vrt, int C, int C4, int H, int W,
             long long N, Gate gate)
{
    if (gated_off(gate)) return;
    const int V = VT > 0 ? VT : Vrt;
    extern __shared__ __align__(16) unsigned char smem[];
    TapRec *recs = reinterpret_cast<TapRec *>(smem);
    float *gtile = reinterpret_cast<float *>(smem + sizeof(TapRec) * kTileVox * V);   // [256 ch][kTileVox + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, cg = blockIdx.z;
    for (long long n0 = (long long)blockIdx.x * kTileVox; n0 < N; n0 += (long long)gridDim.x * kTileVox) {   // see k_fwd_gather
    const long long mapsz = (long long)H * W * C4;

    build_records(recs, proj, coords, b, V, n0, N, H, W, C4);
    {   // grad_out tile, coalesced along voxels: the two half-waves load alternate channels
        const int vl = lane & (kTileVox - 1), half = lane / kTileVox;
        const long long n = n0 + vl;
        for (int r = wave * 64 + half; r < wave * 64 + 64; r += 64 / kTileVox) {
            const int c = cg * kGroupCh + r;
            float g = 0.f;
            if (c < C && n < N) g = to_f32<TO>(grad_out[((long long)b * C + c) * N + n]);
            gtile[r * (kTileVox + 1) + vl] = g;
        }
    }
    __syncthreads();

    int ch[4];
    bool act[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cg * kGroupCh + i * 64 + lane;
        act[i] = c < C;
        ch[i] = act[i] ? c : 0;
    }
    const TF *fb = featT + (long long)b * V * mapsz;
    float *gb = gradT + (long long)b * V * mapsz;

    auto sample4 = [&](const UTap &u, int v, float (&sv)[4]) {
        const TF *fv = fb + v * mapsz;
        if (u.w00 == 0.f && u.w01 == 0.f && u.w10 == 0.f && u.w11 == 0.f) {        // identically zero: reads nothing (see k_fwd_gather)
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = 0.f;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            sv[i] = bilerp(to_f32<TF>(fv[u.o00 + ch[i]]), to_f32<TF>(fv[u.o01 + ch[i]]), to_f32<TF>(fv[u.o10 + ch[i]]),
                           to_f32<TF>(fv[u.o11 + ch[i]]), u.w00, u.w01, u.w10, u.w11);
    };
    auto scatter4 = [&](const UTap &u, int v, const float (&dsv)[4]) {
        float *gv = gb + v * mapsz;
        // zero-weight taps (outside the map, or z <= 0) receive nothing -- wave-uniform branches
        if (u.w00 != 0.f) { for (int i = 0; i < 4; ++i) if (act[i]) atomicAdd(gv + u.o00 + ch[i], dsv[i] * u.w00); }
        if (u.w01 != 0.f) { for (int i = 0; i < 4; ++i) if (act[i]) atomicAdd(gv + u.o01 + ch[i], dsv[i] * u.w01); }
        if (u.w10 != 0.f) { for (int i = 0; i < 4; ++i) if (act[i]) atomicAdd(gv + u.o10 + ch[i], dsv[i] * u.w10); }
        if (u.w11 != 0.f) { for (int i = 0; i < 4; ++i) if (act[i]) atomicAdd(gv + u.o11 + ch[i], dsv[i] * u.w11); }
    };

    for (int jj = 0; jj < kTileVox / 4; ++jj) {
        const int j = wave * (kTileVox / 4) + jj;
        if (n0 + j >= N) break;
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = gtile[(i * 64 + lane) * (kTileVox + 1) + j];

        if constexpr (VT > 0) {
            float s[4][VT], ds[4][VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                float sv[4];
                sample4(uniform_rec(recs[j * VT + v]), v, sv);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = sv[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) aggregate_grad<METHOD, VT>(s[i], g[i], ds[i]);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const float dsv[4] = {ds[0][v], ds[1][v], ds[2][v], ds[3][v]};
                scatter4(uniform_rec(recs[j * VT + v]), v, dsv);
            }
        } else {
            // run-time view count: pass 1 accumulates the aggregate, pass 2 re-samples and scatters
            RunningAgg<METHOD> ra[4];
            int am[4] = {0, 0, 0, 0};
            float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i].init();
            for (int v = 0; v < V; ++v) {
                float sv[4];
                sample4(uniform_rec(recs[j * V + v]), v, sv);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ra[i].push(sv[i]);
                    if (sv[i] > best[i]) { best[i] = sv[i]; am[i] = v; }
                }
            }
            for (int v = 0; v < V; ++v) {
                const UTap u = uniform_rec(recs[j * V + v]);
                float sv[4], dsv[4];
                sample4(u, v, sv);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (METHOD == AGG_SUM) dsv[i] = g[i];
                    else if constexpr (METHOD == AGG_MEAN) dsv[i] = __fdiv_rn(g[i], (float)V);
                    else if constexpr (METHOD == AGG_MAX) dsv[i] = am[i] == v ? g[i] : 0.f;
                    else {
                        const float rden = __builtin_amdgcn_rcpf(ra[i].den);
                        const float o = ra[i].num * rden;
                        dsv[i] = g[i] * __expf(sv[i] - ra[i].m) * rden * (1.f + sv[i] - o);
                    }
                }
                scatter4(u, v, dsv);
            }
        }
    }
    __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ layout passes
// (BV, C, HW) -> (BV, HW, C4): 64 x 64 tiles turned through LDS, both sides coalesced.  VEC: 4 pixels per load and 4 channels per
// store (16-B / 8-B accesses; HW % 4 == 0 and C4 % 4 == 0, base pointers aligned) -- 4.6 -> 5.8 TB/s for the 302 MB of configs[1].
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
k_to_channels_last(const T *__restrict__ src, T *__restrict__ dst, int C, int C4, int HW, Gate gate)
{
    if (gated_off(gate)) return;
    __shared__ T t[64][65];
    const long long bv = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    if constexpr (VEC) {
        struct alignas(sizeof(T) * 4) V4 { T x, y, z, w; };
        const int l16 = threadIdx.x & 15, g = threadIdx.x >> 4;                   // 16 lanes x 4 pixels = one 64-pixel row; 16 rows per pass
        for (int r = g; r < 64; r += 16) {
            const int c = c0 + r, p = p0 + 4 * l16;
            V4 v = {from_f32<T>(0.f), from_f32<T>(0.f), from_f32<T>(0.f), from_f32<T>(0.f)};
            if (c < C && p < HW) v = *reinterpret_cast<const V4 *>(src + (bv * C + c) * HW + p);
            t[r][4 * l16] = v.x; t[r][4 * l16 + 1] = v.y; t[r][4 * l16 + 2] = v.z; t[r][4 * l16 + 3] = v.w;
        }
        __syncthreads();
        for (int r = g; r < 64; r += 16) {
            const int p = p0 + r, c = c0 + 4 * l16;
            if (p < HW && c < C4) {
                const V4 v = {t[4 * l16][r], t[4 * l16 + 1][r], t[4 * l16 + 2][r], t[4 * l16 + 3][r]};
                *reinterpret_cast<V4 *>(dst + (bv * HW + p) * C4 + c) = v;
            }
        }
    } else {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int r = w; r < 64; r += 4) {
            const int c = c0 + r, p = p0 + lane;
            t[r][lane] = (c < C && p < HW) ? src[(bv * C + c) * HW + p] : from_f32<T>(0.f);
        }
        __syncthreads();
        for (int r = w; r < 64; r += 4) {
            const int p = p0 + r, c = c0 + lane;
            if (p < HW && c < C4) dst[(bv * HW + p) * C4 + c] = t[lane][r];
        }
    }
}

// column-major quad-planar fp32 (BV, C/4, W, H, 4) -> channels-last (BV, HW, C) in the feature dtype: what the gather kernels read when
// the caller handed over MVHMR_LAYOUT_QUAD (the fused 1x1 conv's output) and the geometry gate selects them.  One block = one image
// column x, a band of 2^band_log2 rows, every channel quad: (16 << band_log2)-B runs read along y per quad, one C * sizeof(T) run
// written per pixel.
template <typename T>
__global__ void __launch_bounds__(256)
k_quad_to_channels_last(const float4 *__restrict__ src, T *__restrict__ dst, int C, int H, int W, int band_log2, Gate gate)
{
    if (gated_off(gate)) return;
    extern __shared__ float4 qtile[];                                            // [row][quad], quad count padded to odd
    const int nq = C >> 2, ldq = nq | 1, band = 1 << band_log2;
    const long long bv = blockIdx.y;
    const int bands = (H + band - 1) >> band_log2;
    const int x = blockIdx.x / bands, y0 = (blockIdx.x % bands) << band_log2;
    const int rows = H - y0 < band ? H - y0 : band;
    const long long HW = (long long)H * W;
    const float4 *sp = src + bv * nq * HW + (long long)x * H + y0;
    for (int i = threadIdx.x; i < band * nq; i += 256) {
        const int q = i >> band_log2, r = i & (band - 1);
        if (r < rows) qtile[r * ldq + q] = sp[(long long)q * HW + r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * nq; i += 256) {
        const int r = i / nq, q = i - r * nq;
        const float4 v = qtile[r * ldq + q];
        Vec4<T>::store(dst + ((bv * HW + (long long)(y0 + r) * W + x) * C + 4 * q), f32x4{{v.x, v.y, v.z, v.w}});
    }
}

// fp32 channels-last gradient accumulator (BV, HW, C4) -> (BV, C, HW) in the feature dtype
template <typename T>
__global__ void __launch_bounds__(256)
k_grad_to_planar(const float *__restrict__ srcT, T *__restrict__ dst, int C, int C4, int HW, Gate gate)
{
    if (gated_off(gate)) return;
    __shared__ float t[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long bv = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int r = w; r < 64; r += 4) {
        const int p = p0 + r, c = c0 + lane;
        t[r][lane] = (p < HW && c < C4) ? srcT[(bv * HW + p) * C4 + c] : 0.f;
    }
    __syncthreads();
    for (int r = w; r < 64; r += 4) {
        const int c = c0 + r, p = p0 + lane;
        if (c < C && p < HW) dst[(bv * C + c) * HW + p] = from_f32<T>(t[lane][r]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_grad_cast(const float *__restrict__ src, T *__restrict__ dst, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dst[i] = from_f32<T>(src[i]);
}

// coords[b,i,j,k,:] = rot[b] @ (pos + step * (i,j,k) - center[b]) + center[b]   (aggregation.py:150-187)
__global__ void __launch_bounds__(256)
k_build_coords(float *__restrict__ coords, const float *__restrict__ rot, const float *__restrict__ center, int S,
               float px, float py, float pz, float sx, float sy, float sz)
{
    const int b = blockIdx.y;
    const long long N = (long long)S * S * S;
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int k = (int)(n % S), j = (int)((n / S) % S), i = (int)(n / ((long long)S * S));
    const float *R = rot + b * 9, *ce = center + b * 3;
    // grid_coord = position + (sides / (S-1)) * grid   (fp32 mul then add, aggregation.py:157-159)
    const float gx = __fadd_rn(px, __fmul_rn(sx, (float)i));
    const float gy = __fadd_rn(py, __fmul_rn(sy, (float)j));
    const float gz = __fadd_rn(pz, __fmul_rn(sz, (float)k));
    const float dx = __fsub_rn(gx, ce[0]), dy = __fsub_rn(gy, ce[1]), dz = __fsub_rn(gz, ce[2]);   // :184
    float *o = coords + ((long long)b * N + n) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) {                                                                      // :185, volumetric.py:110
        const float acc = __fmaf_rn(R[r * 3 + 2], dz, __fmaf_rn(R[r * 3 + 1], dy, __fmul_rn(R[r * 3 + 0], dx)));
        o[r] = __fadd_rn(acc, ce[r]);                                                                  // :186
    }
}

// ------------------------------------------------------------------------------------------ launchers
template <typename TF, typename TO, int METHOD>
static hipError_t fwd_dispatch_v(const TF *featT, const float *proj, const Coords &coords, TO *out, const Problem &p,
                                 hipStream_t s)
{
    const int Q = p.C4 / 4;
    int tstride = (Q < kGroupQuads ? Q : kGroupQuads) + 1;
    tstride |= 1;
    const size_t lds = sizeof(TapRec) * kTileVox * (size_t)p.V + sizeof(f32x4) * kTileVox * (size_t)tstride;
    const unsigned tiles = (unsigned)((p.N + kTileVox - 1) / kTileVox);
    const dim3 grid(p.gate_count && tiles > kGatedTiles ? kGatedTiles : tiles, (unsigned)p.B, (unsigned)((Q + kGroupQuads - 1) / kGroupQuads));
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, featT, proj, coords, out, p.V, p.C, p.C4, p.H, p.W, p.N, tstride, make_gate(p, false));
        return hipGetLastError();
    };
    switch (p.V) {
    case 2: return go(k_fwd_gather<TF, TO, METHOD, 2>);
    case 4: return go(k_fwd_gather<TF, TO, METHOD, 4>);
    case 8: return go(k_fwd_gather<TF, TO, METHOD, 8>);
    default: return go(k_fwd_gather<TF, TO, METHOD, 0>);
    }
}

template <typename TF, typename TO>
static hipError_t fwd_dispatch_m(const TF *featT, const float *proj, const Coords &coords, TO *out, const Problem &p,
                                 hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return fwd_dispatch_v<TF, TO, AGG_SOFTMAX>(featT, proj, coords, out, p, s);
    case AGG_SUM: return fwd_dispatch_v<TF, TO, AGG_SUM>(featT, proj, coords, out, p, s);
    case AGG_MEAN: return fwd_dispatch_v<TF, TO, AGG_MEAN>(featT, proj, coords, out, p, s);
    case AGG_MAX: return fwd_dispatch_v<TF, TO, AGG_MAX>(featT, proj, coords, out, p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fwd_gather(const void *featT, const float *proj, const Coords &coords, void *out, const Problem &p,
                             hipStream_t s)
{
    if (p.out_bf16) return p.feat_f16 ? hipErrorNotSupported : fwd_dispatch_m((const float *)featT, proj, coords, (bf16_t *)out, p, s);
    if (!p.feat_f16 && !p.out_f16) return fwd_dispatch_m((const float *)featT, proj, coords, (float *)out, p, s);
    if (p.feat_f16 && p.out_f16) return fwd_dispatch_m((const __half *)featT, proj, coords, (__half *)out, p, s);
    if (p.feat_f16 && !p.out_f16) return fwd_dispatch_m((const __half *)featT, proj, coords, (float *)out, p, s);
    return hipErrorNotSupported;
}

template <typename TF, typename TO, int METHOD>
static hipError_t bwd_dispatch_v(const TO *go_, const TF *featT, const float *proj, const Coords &coords, float *gradT,
                                 const Problem &p, hipStream_t s)
{
    const size_t lds = sizeof(TapRec) * kTileVox * (size_t)p.V + sizeof(float) * kGroupCh * (kTileVox + 1);
    const unsigned tiles = (unsigned)((p.N + kTileVox - 1) / kTileVox);
    const dim3 grid(p.gate_count && tiles > kGatedTiles ? kGatedTiles : tiles, (unsigned)p.B, (unsigned)((p.C + kGroupCh - 1) / kGroupCh));
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, go_, featT, proj, coords, gradT, p.V, p.C, p.C4, p.H, p.W, p.N, make_gate(p, false));
        return hipGetLastError();
    };
    switch (p.V) {
    case 2: return go(k_bwd_gather<TF, TO, METHOD, 2>);
    case 4: return go(k_bwd_gather<TF, TO, METHOD, 4>);
    case 8: return go(k_bwd_gather<TF, TO, METHOD, 8>);
    default: return go(k_bwd_gather<TF, TO, METHOD, 0>);
    }
}

template <typename TF, typename TO>
static hipError_t bwd_dispatch_m(const TO *go_, const TF *featT, const float *proj, const Coords &coords, float *gradT,
                                 const Problem &p, hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return bwd_dispatch_v<TF, TO, AGG_SOFTMAX>(go_, featT, proj, coords, gradT, p, s);
    case AGG_SUM: return bwd_dispatch_v<TF, TO, AGG_SUM>(go_, featT, proj, coords, gradT, p, s);
    case AGG_MEAN: return bwd_dispatch_v<TF, TO, AGG_MEAN>(go_, featT, proj, coords, gradT, p, s);
    case AGG_MAX: return bwd_dispatch_v<TF, TO, AGG_MAX>(go_, featT, proj, coords, gradT, p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_bwd_gather(const void *grad_out, const void *featT, const float *proj, const Coords &coords,
                             float *gradT, const Problem &p, hipStream_t s)
{
    if (p.out_bf16) return p.feat_f16 ? hipErrorNotSupported : bwd_dispatch_m((const bf16_t *)grad_out, (const float *)featT, proj, coords, gradT, p, s);
    if (!p.feat_f16 && !p.out_f16) return bwd_dispatch_m((const float *)grad_out, (const float *)featT, proj, coords, gradT, p, s);
    if (p.feat_f16 && p.out_f16) return bwd_dispatch_m((const __half *)grad_out, (const __half *)featT, proj, coords, gradT, p, s);
    if (p.feat_f16 && !p.out_f16) return bwd_dispatch_m((const float *)grad_out, (const __half *)featT, proj, coords, gradT, p, s);
    return hipErrorNotSupported;
}

hipError_t launch_to_channels_last(const void *src, void *dst, const Problem &p, hipStream_t s)
{
    const int HW = p.H * p.W;
    const dim3 grid((HW + 63) / 64, (p.C4 + 63) / 64, (unsigned)(p.B * p.V));
    const bool vec = HW % 4 == 0 && p.C4 % 4 == 0 && (reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) % 16 == 0;
    const Gate gate = make_gate(p, false);
    if (p.feat_f16) {
        if (vec) hipLaunchKernelGGL((k_to_channels_last<__half, true>), grid, dim3(256), 0, s, (const __half *)src, (__half *)dst, p.C, p.C4, HW, gate);
        else hipLaunchKernelGGL((k_to_channels_last<__half, false>), grid, dim3(256), 0, s, (const __half *)src, (__half *)dst, p.C, p.C4, HW, gate);
    } else {
        if (vec) hipLaunchKernelGGL((k_to_channels_last<float, true>), grid, dim3(256), 0, s, (const float *)src, (float *)dst, p.C, p.C4, HW, gate);
        else hipLaunchKernelGGL((k_to_channels_last<float, false>), grid, dim3(256), 0, s, (const float *)src, (float *)dst, p.C, p.C4, HW, gate);
    }
    return hipGetLastError();
}

// rows per block: the largest power of two (<= 32) whose tile of all channel quads fits 64 KiB of LDS; -1: not even 4 rows (C > 4092)
static int quad_band_log2(const Problem &p)
{
    for (int b = 5; b >= 2; --b)
        if (((size_t)((p.C / 4) | 1) * sizeof(float4) << b) <= 64 * 1024) return b;
    return -1;
}

bool quad_to_channels_last_supported(const Problem &p)
{
    return p.C % 4 == 0 && quad_band_log2(p) >= 0 && (long long)p.B * p.V <= 65535;
}

hipError_t launch_quad_to_channels_last(const void *srcK, void *dst, const Problem &p, hipStream_t s)
{
    if (!quad_to_channels_last_supported(p)) return hipErrorNotSupported;
    const int b = quad_band_log2(p);
    const size_t lds = (size_t)((p.C / 4) | 1) * sizeof(float4) << b;
    const dim3 grid((unsigned)(p.W * ((p.H + (1 << b) - 1) >> b)), (unsigned)(p.B * p.V));
    if (p.feat_f16) hipLaunchKernelGGL(k_quad_to_channels_last<__half>, grid, dim3(256), lds, s, (const float4 *)srcK, (__half *)dst, p.C, p.H, p.W, b, make_gate(p, false));
    else hipLaunchKernelGGL(k_quad_to_channels_last<float>, grid, dim3(256), lds, s, (const float4 *)srcK, (float *)dst, p.C, p.H, p.W, b, make_gate(p, false));
    return hipGetLastError();
}

hipError_t launch_grad_to_planar(const float *srcT, void *dst, const Problem &p, hipStream_t s)
{
    const int HW = p.H * p.W;
    const dim3 grid((HW + 63) / 64, (p.C4 + 63) / 64, (unsigned)(p.B * p.V));
    if (p.feat_f16) hipLaunchKernelGGL(k_grad_to_planar<__half>, grid, dim3(256), 0, s, srcT, (__half *)dst, p.C, p.C4, HW, make_gate(p, false));
    else hipLaunchKernelGGL(k_grad_to_planar<float>, grid, dim3(256), 0, s, srcT, (float *)dst, p.C, p.C4, HW, make_gate(p, false));
    return hipGetLastError();
}

hipError_t launch_grad_cast(const float *srcT, void *dst, const Problem &p, hipStream_t s)
{
    const long long n = (long long)p.B * p.V * p.H * p.W * p.C4;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (p.feat_f16) hipLaunchKernelGGL(k_grad_cast<__half>, dim3(blocks), dim3(256), 0, s, srcT, (__half *)dst, n);
    else hipLaunchKernelGGL(k_grad_cast<float>, dim3(blocks), dim3(256), 0, s, srcT, (float *)dst, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ DLT triangulation
// One point per sample from V views (utils/multiview.py:112-168, Hartley & Zisserman 12.2): the right singular vector of the smallest
// singular value of A (2V x 4), rows u * P[2,:] - P[0,:] and v * P[2,:] - P[1,:] -- with inconsistent rays (the image centres of a
// camera ring do not meet in one point) that is the minimiser of |A h| over UNIT h, so A must not be rescaled column-wise: the
// constraint would change and with it the answer.  One thread per sample, float64 throughout: the smallest eigenvector of the 4 x 4
// normal matrix A^T A by cyclic Jacobi rotations -- no SVD, no host round trip.  cond(A) is ~10^1..10^4 for pixel-scale projection
// matrices, its square far inside float64 (agreement with numpy's float64 SVD: tests/test_unproject_gpu.py).
__global__ void __launch_bounds__(64)
k_triangulate_dlt(const float *__restrict__ proj, const float *__restrict__ points, const float *__restrict__ conf, float *__restrict__ out, int B, int V,
                  int points_per_sample, int conf_per_sample)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float *P = proj + (long long)b * V * 12;
    const float *uv = points + (points_per_sample ? (long long)b * V * 2 : 0);
    double M[4][4] = {{0}}, E[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    const float *cf = conf ? conf + (conf_per_sample ? (long long)b * V : 0) : nullptr;    // A *= confidences (utils/multiview.py:156-161): rows of view v times c_v
    for (int v = 0; v < V; ++v)
        for (int r = 0; r < 2; ++r) {
            double a[4];
            const double w = cf ? (double)cf[v] : 1.0;
            for (int k = 0; k < 4; ++k) a[k] = w * ((double)uv[2 * v + r] * (double)P[v * 12 + 8 + k] - (double)P[v * 12 + 4 * r + k]);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) M[i][j] += a[i] * a[j];
        }
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) off += M[i][j] * M[i][j];
        if (off == 0.0) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (M[p][q] == 0.0) continue;
                const double th = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 4; ++k) {                                     // M <- M J
                    const double mkp = M[k][p], mkq = M[k][q];
                    M[k][p] = c * mkp - sn * mkq; M[k][q] = sn * mkp + c * mkq;
                }
                for (int k = 0; k < 4; ++k) {                                     // M <- J^T M, E <- E J
                    const double mpk = M[p][k], mqk = M[q][k];
                    M[p][k] = c * mpk - sn * mqk; M[q][k] = sn * mpk + c * mqk;
                    const double ekp = E[k][p], ekq = E[k][q];
                    E[k][p] = c * ekp - sn * ekq; E[k][q] = sn * ekp + c * ekq;
                }
            }
    }
    int m = 0;
    for (int k = 1; k < 4; ++k) m = M[k][k] < M[m][m] ? k : m;
    const double h0 = E[0][m], h1 = E[1][m], h2 = E[2][m], h3 = E[3][m];
    out[3 * b + 0] = (float)(h0 / h3); out[3 * b + 1] = (float)(h1 / h3); out[3 * b + 2] = (float)(h2 / h3);
}

hipError_t launch_triangulate_dlt(const float *proj, const float *points, const float *conf, float *out, int B, int V, int points_per_sample,
                                  int conf_per_sample, hipStream_t s)
{
    hipLaunchKernelGGL(k_triangulate_dlt, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, proj, points, conf, out, B, V, points_per_sample, conf_per_sample);
    return hipGetLastError();
}

hipError_t launch_build_coords(float *coords, const float *rot, const float *center, int B, int S, const double pos[3],
                               const double sides[3], hipStream_t s)
{
    const long long N = (long long)S * S * S;
    const double d = (double)(S - 1);
    // step = sides / (S-1) and position are float64 scalars in the reference, cast to fp32 when they meet the tensor
    const float sx = (float)(sides[0] / d), sy = (float)(sides[1] / d), sz = (float)(sides[2] / d);
    hipLaunchKernelGGL(k_build_coords, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, s, coords, rot, center, S,
                       (float)pos[0], (float)pos[1], (float)pos[2], sx, sy, sz);
    return hipGetLastError();
}

}  // namespace mvhmr
