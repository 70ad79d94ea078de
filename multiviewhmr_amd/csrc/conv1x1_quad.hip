// process_feature fused with the layout pass (reference models/aggregation.py:108-110,189-191):
//     y[bv, co, p] = sum_ci W[co, ci] * x[bv, ci, p] + bias[co]
// as an fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the vector rate, MI355X_MICROARCH.md) whose epilogue
// writes the COLUMN-major quad-planar copy (B,V,Cout/4,Wf,Hf,4) that the brick forward stages -- the (B,V,Cout,Hf,Wf) tensor
// the reference's nn.Conv2d produces, and the layout pass over it, never exist.
//
// Mapping (CDNA4, wave64).  Block = 256 threads, output tile = 128 channels x 128 pixels (4 image rows x 32 columns); the four
// waves own 64 x 64 sub-tiles = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers per lane).  K (= Cin) is staged through LDS
// in chunks of 16, double-buffered: A = W tile [128][16] (row stride 17: conflict-free fragment reads), B = x tile [16][4 rows]
// [32 + 8 pad] (the pad makes the (row, column) -> MFMA column map below conflict-free).  Pixel j of a 32-wide MFMA tile is
// (row j % 4, column j / 4) of an 8-column strip, so that the four lanes j .. j+3 hold four consecutive ROWS of one image column:
// in the column-major destination they are 64 contiguous bytes, one TCP access per lane quad.
// The 32x32x2 accumulator layout gives every lane 4 consecutive output channels of one pixel per register group: exactly one
// 16-B slot of the quad-planar layout, so the epilogue is four float4 stores per MFMA tile, no shuffles.
#include "device_common.h"
#include "kernels.h"

namespace mvhmr {

namespace {
constexpr int kTM = 128, kTN = 128, kTK = 16;
constexpr int kLdA = kTK + 1;                  // 17: lanes of a fragment read (consecutive rows) hit distinct banks
constexpr int kRowB = 32 + 8;                  // image-row stride inside a k-row of B: bank = 8 * (j % 4) + j / 4, all 32 distinct
constexpr int kLdB = 4 * kRowB;                // 160 floats per k
typedef float f32x16 __attribute__((ext_vector_type(16)));
}  // namespace

// QUAD_OUT = false is the same GEMM with a planar result (BV, Cout, H*W): the tile is 128 consecutive pixels of the flattened map (MFMA
// column j = pixel j), the epilogue writes 128-B runs per channel.  Used for the gradient w.r.t. the conv input (weights transposed
// by the caller), where hipBLASLt's batched pick runs at half this kernel's rate.
// requires Cin % 16 == 0, Cout % 128 == 0 and (QUAD_OUT: H % 4 == 0, W % 32 == 0; planar: H * W % 128 == 0) (checked by the launchers).
// Pinned to 4 waves per SIMD (accumulators in VGPRs, 102 registers): at 3 the MFMA pipe is 65 % busy, at 4 ~82 %.
template <bool QUAD_OUT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_conv1x1_quad(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, float4 *__restrict__ dst,
               int Cin, int Cout, int H, int W)
{
    __shared__ float sA[2][kTM * kLdA];
    __shared__ float sB[2][kTK * kLdB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = W >> 5;
    const int x0 = QUAD_OUT ? (blockIdx.x % tiles_x) << 5 : 0, y0 = QUAD_OUT ? (blockIdx.x / tiles_x) << 2 : 0;
    const int p0 = blockIdx.x * kTN;                                             // planar form: first pixel of the tile
    const int m0 = blockIdx.y * kTM;
    const long long bv = blockIdx.z;
    const long long HW = (long long)H * W;
    const float *xb = x + bv * Cin * HW;

    // global -> LDS assignments: A: thread t loads W[m0 + t/2][k0 + 8*(t%2) .. +8]; B: thread t loads x[k0 + t/16][y0 + (t%16)/4][x0 + 8*(t%4) .. +8]
    const int a_row = tid >> 1, a_k = (tid & 1) << 3;
    const int b_k = tid >> 4, b_y = (tid & 15) >> 2, b_x = (tid & 3) << 3;
    const float *ga = w + (long long)(m0 + a_row) * Cin + a_k;
    // (planar form: the same thread -> LDS map with "image row" b_y = 32-pixel segment b_y of the tile)
    const float *gb = QUAD_OUT ? xb + (long long)b_k * HW + (long long)(y0 + b_y) * W + x0 + b_x
                               : xb + (long long)b_k * HW + p0 + b_y * 32 + b_x;
    float4 ra0, ra1, rb0, rb1;                                                   // scalars, not arrays: they must stay in registers
    auto gload = [&](int k0) __attribute__((always_inline)) {
        ra0 = *reinterpret_cast<const float4 *>(ga + k0); ra1 = *reinterpret_cast<const float4 *>(ga + k0 + 4);
        const float *p = gb + (long long)k0 * HW;
        rb0 = *reinterpret_cast<const float4 *>(p); rb1 = *reinterpret_cast<const float4 *>(p + 4);
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        float *a = &sA[buf][a_row * kLdA + a_k];
        a[0] = ra0.x; a[1] = ra0.y; a[2] = ra0.z; a[3] = ra0.w; a[4] = ra1.x; a[5] = ra1.y; a[6] = ra1.z; a[7] = ra1.w;
        float *b = &sB[buf][b_k * kLdB + b_y * kRowB + b_x];
        *reinterpret_cast<float4 *>(b) = rb0;
        *reinterpret_cast<float4 *>(b + 4) = rb1;
    };

    // this wave's 64 x 64 sub-tile: rows wm .. wm+63 of the block tile, MFMA column tiles 2*wn, 2*wn+1 (8 image columns each)
    const int wm = (wave >> 1) << 6, wn = (wave & 1) << 1;
    const int fi = lane & 31, fk = lane >> 5;                                    // fragment row / column index, k within the pair
    // LDS offset of MFMA column fi of column tile c: quad form (row fi%4, column 8c + fi/4); planar form pixel 32c + fi = segment c, offset fi
    const int pj = QUAD_OUT ? (fi & 3) * kRowB + (fi >> 2) : fi;
    constexpr int kTileStep = QUAD_OUT ? 8 : kRowB;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = Cin / kTK;
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) gload((kc + 1) * kTK);                                   // next chunk in flight under the MFMAs
        const float *A = sA[buf], *Bm = sB[buf];
#pragma unroll
        for (int k = 0; k < kTK; k += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = A[(wm + 32 * i + fi) * kLdA + k + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bm[(k + fk) * kLdB + (wn + j) * kTileStep + pj];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: accumulator register r of lane (fi, fk) is output row 8*(r/4) + 4*fk + r%4, column fi of the MFMA tile
    if constexpr (QUAD_OUT) {
        const int Q = Cout >> 2;
        const int py = y0 + (fi & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = x0 + (wn + j) * 8 + (fi >> 2);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = m0 + wm + 32 * i + 8 * g + 4 * fk;               // first of 4 consecutive channels
                    float4 o = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    if (bias) { const float4 bq = *reinterpret_cast<const float4 *>(bias + co); o.x += bq.x; o.y += bq.y; o.z += bq.z; o.w += bq.w; }
                    dst[(bv * Q + (co >> 2)) * HW + (long long)px * H + py] = o;
                }
            }
    } else {
        float *out = reinterpret_cast<float *>(dst) + bv * Cout * HW;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int px = p0 + (wn + j) * 32 + fi;                           // 32 lanes = 128 contiguous bytes of one channel
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * fk + (r & 3);
                    out[(long long)co * HW + px] = acc[i][j][r] + (bias ? bias[co] : 0.f);
                }
            }
    }
}

bool conv1x1_quad_supported(int Cin, int Cout, int H, int W) { return Cin % kTK == 0 && Cout % kTM == 0 && H % 4 == 0 && W % 32 == 0; }
bool conv1x1_planar_supported(int Cin, int Cout, int HW) { return Cin % kTK == 0 && Cout % kTM == 0 && HW % kTN == 0; }

hipError_t launch_conv1x1_quad(const float *x, const float *w, const float *bias, void *dst, int BV, int Cin, int Cout, int H, int W,
                               hipStream_t s)
{
    if (!conv1x1_quad_supported(Cin, Cout, H, W)) return hipErrorNotSupported;
    const dim3 grid((W / 32) * (H / 4), Cout / kTM, BV);
    hipLaunchKernelGGL(k_conv1x1_quad<true>, grid, dim3(256), 0, s, x, w, bias, (float4 *)dst, Cin, Cout, H, W);
    return hipGetLastError();
}

// y[bv, co, p] = sum_ci w[co, ci] * x[bv, ci, p] (+ bias[co]), everything planar fp32
hipError_t launch_conv1x1_planar(const float *x, const float *w, const float *bias, float *dst, int BV, int Cin, int Cout, int HW, hipStream_t s)
{
    if (!conv1x1_planar_supported(Cin, Cout, HW)) return hipErrorNotSupported;
    const dim3 grid(HW / kTN, Cout / kTM, BV);
    hipLaunchKernelGGL(k_conv1x1_quad<false>, grid, dim3(256), 0, s, x, w, bias, (float4 *)dst, Cin, Cout, 1, HW);
    return hipGetLastError();
}

// Weight and bias gradient of the 1x1 conv:  dW[co, ci] += sum_{bv, p} gy[bv, co, p] * x[bv, ci, p],  db[co] += sum gy[bv, co, p]
// (autograd through models/aggregation.py:189-191).  A GEMM with M = Cout, N = Cin and K = BV * H * W (1.2 M at the north-star size):
// split along K over the grid -- a block owns a 128 x 128 tile of dW and one slice of one map's pixels, stages 32-pixel chunks of both
// operands through LDS ([row][k], stride 33: conflict-free fragment reads), keeps the next chunk in registers while the MFMAs run, and
// adds its partial tile with float atomics shaped as 128 contiguous bytes per wave row (dW / db are zeroed by the caller).
namespace {
constexpr int kWK = 32;                         // pixels per chunk
constexpr int kWLd = kWK + 1;
}  // namespace

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_conv1x1_wgrad(const float *__restrict__ gy, const float *__restrict__ x, float *__restrict__ dW, float *__restrict__ db,
                int Cout, int Cin, int HW, int slices_per_map, int slice_px)
{
    __shared__ float sA[kTM * kWLd];
    __shared__ float sB[kTN * kWLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bv = blockIdx.x / slices_per_map, p_begin = (blockIdx.x % slices_per_map) * slice_px;
    const int m0 = blockIdx.y * kTM, n0 = blockIdx.z * kTN;
    // global -> LDS: thread t stages 16 consecutive pixels of row t / 2 of both tiles
    const int row = tid >> 1, kp = (tid & 1) << 4;
    const float *ga = gy + ((long long)bv * Cout + m0 + row) * HW + p_begin + kp;
    const float *gb = x + ((long long)bv * Cin + n0 + row) * HW + p_begin + kp;
    float4 a0, a1, a2, a3, b0, b1, b2, b3;
    float bsum = 0.f;
    auto gload = [&](int k0) __attribute__((always_inline)) {
        const float4 *pa = reinterpret_cast<const float4 *>(ga + k0), *pb = reinterpret_cast<const float4 *>(gb + k0);
        a0 = pa[0]; a1 = pa[1]; a2 = pa[2]; a3 = pa[3];
        b0 = pb[0]; b1 = pb[1]; b2 = pb[2]; b3 = pb[3];
    };
    auto put4 = [&](float *d, const float4 &v) __attribute__((always_inline)) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
    auto lstore = [&]() __attribute__((always_inline)) {
        float *a = &sA[row * kWLd + kp], *b = &sB[row * kWLd + kp];
        put4(a, a0); put4(a + 4, a1); put4(a + 8, a2); put4(a + 12, a3);
        put4(b, b0); put4(b + 4, b1); put4(b + 8, b2); put4(b + 12, b3);
        bsum += ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w)) + ((a2.x + a2.y) + (a2.z + a2.w)) + ((a3.x + a3.y) + (a3.z + a3.w));
    };
    const int wm = (wave >> 1) << 6, wn = (wave & 1) << 6;
    const int fi = lane & 31, fk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = slice_px / kWK;
    gload(0);
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();                                                         // every wave has read the previous chunk
        lstore();
        __syncthreads();
        if (kc + 1 < nk) gload((kc + 1) * kWK);                                   // in flight under the MFMAs
#pragma unroll
        for (int k = 0; k < kWK; k += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = sA[(wm + 32 * i + fi) * kWLd + k + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = sB[(wn + 32 * j + fi) * kWLd + k + fk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // accumulator register r of lane (fi, fk): row 8*(r/4) + 4*fk + r%4 (co), column fi (ci)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * fk + (r & 3);
                atomicAdd(dW + (long long)co * Cin + n0 + wn + 32 * j + fi, acc[i][j][r]);
            }
    if (db && blockIdx.z == 0) {
        bsum += __shfl_xor(bsum, 1);
        if ((tid & 1) == 0) atomicAdd(db + m0 + row, bsum);
    }
}

bool conv1x1_wgrad_supported(int Cin, int Cout, int HW) { return Cin % kTN == 0 && Cout % kTM == 0 && HW % kWK == 0; }

// dW (Cout, Cin) and db (Cout, may be null) are ADDED INTO: zero them first
hipError_t launch_conv1x1_wgrad(const float *gy, const float *x, float *dW, float *db, int BV, int Cin, int Cout, int HW, hipStream_t s)
{
    if (!conv1x1_wgrad_supported(Cin, Cout, HW)) return hipErrorNotSupported;
    // slices per map: enough blocks to fill the chip a few times, slices of whole chunks
    const int chunks = HW / kWK, tiles = (Cout / kTM) * (Cin / kTN);
    int spm = 1;
    for (int c = 1; c <= chunks; ++c)
        if (chunks % c == 0 && (long long)BV * c * tiles <= 2048) spm = c;
    const dim3 grid(BV * spm, Cout / kTM, Cin / kTN);
    hipLaunchKernelGGL(k_conv1x1_wgrad, grid, dim3(256), 0, s, gy, x, dW, db, Cout, Cin, HW, spm, (chunks / spm) * kWK);
    return hipGetLastError();
}

}  // namespace mvhmr
