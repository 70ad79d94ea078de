// Device-side building blocks shared by every un-projection kernel (gfx950 / CDNA4 only).
//
// Arithmetic contract.  The per-view sample must agree with the reference's CPU path
// (models/aggregation.py:38-62 -> ATen sgemm + grid_sampler_2d) as closely as fp32 allows, because
// the projection is ill-conditioned (mm-scale coordinates times ~1e2 px/m focal lengths): a different
// rounding ORDER alone moves the result by ~4e-5 against a 1e-4 bar.  The orders below were pinned
// bit-for-bit against torch 2.10 in the build container (see oracle/unproject_oracle.c):
//   projection   r_k = fma(P_k3, 1, fma(P_k2, X2, fma(P_k1, X1, P_k0 * X0)))           multiview.py:105
//   divide       IEEE (u = a / z), then 2 * (u / Hf - 0.5), ((g + 1) / 2) * (size - 1)  aggregation.py:49-50
//   bilinear     fma(se_v, se, fma(sw_v, sw, fma(ne_v, ne, nw_v * nw)))                 grid_sampler_2d
// Everything here is compiled with -ffp-contract=off; fused operations are spelled out.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "gate.h"

namespace mvhmr {

// Geometry gate (kernels.h): true when this launch is the variant the device-side brick count did NOT select.
__device__ __forceinline__ bool gated_off(const Gate &g)
{
    if (!g.count) return false;
    const bool brick = __builtin_nontemporal_load(g.count) <= g.limit;
    return brick != (g.wants_brick != 0);
}


// voxel centre of (sample b, flat index n = (i*Y + j)*Z + k): tensor read or cuboid recipe (gate.h::Coords)
__device__ __forceinline__ void voxel_xyz(const Coords &cs, int b, long long N, long long n, float &x, float &y, float &z)
{
    if (cs.ptr) {
        const float *X = cs.ptr + ((long long)b * N + n) * 3;
        x = X[0]; y = X[1]; z = X[2];
        return;
    }
    const int k = (int)(n % cs.Z), j = (int)((n / cs.Z) % cs.Y), i = (int)(n / ((long long)cs.Y * cs.Z));
    const float *R = cs.rot + b * 9, *ce = cs.center + b * 3;
    // grid_coord = position + (sides / (S-1)) * grid   (fp32 mul then add, aggregation.py:157-159)
    const float gx = __fadd_rn(cs.px, __fmul_rn(cs.sx, (float)i));
    const float gy = __fadd_rn(cs.py, __fmul_rn(cs.sy, (float)j));
    const float gz = __fadd_rn(cs.pz, __fmul_rn(cs.sz, (float)k));
    const float dx = __fsub_rn(gx, ce[0]), dy = __fsub_rn(gy, ce[1]), dz = __fsub_rn(gz, ce[2]);             // :184
    x = __fadd_rn(__fmaf_rn(R[2], dz, __fmaf_rn(R[1], dy, __fmul_rn(R[0], dx))), ce[0]);                    // :185-186, volumetric.py:110
    y = __fadd_rn(__fmaf_rn(R[5], dz, __fmaf_rn(R[4], dy, __fmul_rn(R[3], dx))), ce[1]);
    z = __fadd_rn(__fmaf_rn(R[8], dz, __fmaf_rn(R[7], dy, __fmul_rn(R[6], dx))), ce[2]);
}

enum : int { AGG_SOFTMAX = 0, AGG_SUM = 1, AGG_MEAN = 2, AGG_MAX = 3 };

// 1 = the transcendentals of the forward's softmax are issued in runs (aggregate2); 0 = wherever the scheduler puts them
// (the backward's aggregate_grad gains nothing from it: 12.98 vs 12.92 ms per call, round 4)
#ifndef MVHMR_AGG_GROUP
#define MVHMR_AGG_GROUP 1
#endif

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kMaxViews = 16;   // per-voxel view records kept in LDS / registers

// One voxel seen by one camera: pixel coordinates of the nw tap and the four bilinear weights.
// A tap outside the map keeps weight 0 and a clamped (always readable) pixel, which is what
// zero padding means for a linear sampler (aggregation.py:55-58); z <= 0 zeroes all four
// (aggregation.py:42,62).  `any` is 0 when the whole sample is exactly zero.
struct Taps {
    int x0, y0, x1, y1;          // clamped to [0, W-1] / [0, H-1]
    float w00, w01, w10, w11;    // (y0,x0) (y0,x1) (y1,x0) (y1,x1) = nw, ne, sw, se
    int rx0, ry0;                // unclamped nw tap, in [-1, W-1] x [-1, H-1] (0,0 when any == 0)
    int any;
};

__device__ __forceinline__ Taps make_taps(const float *__restrict__ P, float X0, float X1, float X2, int H, int W)
{
    const float a = __fmaf_rn(P[3], 1.f, __fmaf_rn(P[2], X2, __fmaf_rn(P[1], X1, __fmul_rn(P[0], X0))));
    const float b = __fmaf_rn(P[7], 1.f, __fmaf_rn(P[6], X2, __fmaf_rn(P[5], X1, __fmul_rn(P[4], X0))));
    const float z = __fmaf_rn(P[11], 1.f, __fmaf_rn(P[10], X2, __fmaf_rn(P[9], X1, __fmul_rn(P[8], X0))));
    Taps t;
    t.x0 = t.y0 = t.x1 = t.y1 = 0;
    t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
    t.rx0 = t.ry0 = 0;
    t.any = 0;
    if (!(z > 0.f)) return t;                        // z <= 0 (or NaN): sample is exactly 0
    const float u = __fdiv_rn(a, z), v = __fdiv_rn(b, z);
    // quirk Q1: x is normalised by feature_shape[0] = Hf, y by feature_shape[1] = Wf
    const float gx = __fmul_rn(2.f, __fsub_rn(__fdiv_rn(u, (float)H), 0.5f));
    const float gy = __fmul_rn(2.f, __fsub_rn(__fdiv_rn(v, (float)W), 0.5f));
    const float ix = __fmul_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f), (float)(W - 1));   // (g+1)/2 is exact either way
    const float iy = __fmul_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f), (float)(H - 1));
    if (!(ix > -1.f && ix < (float)W && iy > -1.f && iy < (float)H)) return t;        // all four taps outside
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.f), ix);
    const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.f), iy);
    const bool xin0 = x0 >= 0, xin1 = x1 <= W - 1, yin0 = y0 >= 0, yin1 = y1 <= H - 1;
    t.w00 = (xin0 && yin0) ? __fmul_rn(wx0, wy0) : 0.f;
    t.w01 = (xin1 && yin0) ? __fmul_rn(wx1, wy0) : 0.f;
    t.w10 = (xin0 && yin1) ? __fmul_rn(wx0, wy1) : 0.f;
    t.w11 = (xin1 && yin1) ? __fmul_rn(wx1, wy1) : 0.f;
    t.x0 = xin0 ? x0 : 0;
    t.y0 = yin0 ? y0 : 0;
    t.x1 = xin1 ? x1 : W - 1;
    t.y1 = yin1 ? y1 : H - 1;
    t.rx0 = x0;
    t.ry0 = y0;
    t.any = 1;
    return t;
}

__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, float w00, float w01, float w10, float w11)
{
    return __fmaf_rn(v11, w11, __fmaf_rn(v10, w10, __fmaf_rn(v01, w01, __fmul_rn(v00, w00))));
}

__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---------------------------------------------------------------- cross-view aggregate (aggregation.py:71-85)
// s[0..V) are the per-view samples of one (voxel, channel); returns the aggregated value.
template <int METHOD, int V>
__device__ __forceinline__ float aggregate(const float (&s)[V])
{
    if constexpr (METHOD == AGG_SUM || METHOD == AGG_MEAN) {
        float r = s[0];
#pragma unroll
        for (int v = 1; v < V; ++v) r = __fadd_rn(r, s[v]);
        return METHOD == AGG_MEAN ? __fdiv_rn(r, (float)V) : r;
    } else if constexpr (METHOD == AGG_MAX) {
        float r = s[0];
#pragma unroll
        for (int v = 1; v < V; ++v) r = fmaxf(r, s[v]);
        return r;
    } else {
        // max without the canonicalising v_max x,x hipcc puts in front of fmaxf (the samples are never signalling NaNs
        // that anything downstream could observe), v_max3 where three operands are at hand
        float m = s[0];
        if constexpr (V >= 3) {
            m = vmax3(s[0], s[1], s[2]);
#pragma unroll
            for (int v = 3; v + 1 < V; v += 2) m = vmax3(m, s[v], s[v + 1]);
            if constexpr ((V - 3) % 2 == 1) m = vmax(m, s[V - 1]);
        } else if constexpr (V == 2) {
            m = vmax(s[0], s[1]);
        }
        // e_v = exp(s_v - m) as exp2((s_v - m) * log2e).  The difference is taken FIRST (exact for the maximum, a few ulps of the
        // difference otherwise): folding it into fma(s_v, log2e, -m * log2e) saves one instruction per view but leaves the
        // rounding of m * log2e in the exponent -- |m| * 2^-24, i.e. weights off by 1e-7 |m| and garbage beyond |m| ~ 1e9
        const float e0 = __builtin_amdgcn_exp2f((s[0] - m) * 1.4426950408889634f);
        float den = e0, num = e0 * s[0];
#pragma unroll
        for (int v = 1; v < V; ++v) {
            const float e = __builtin_amdgcn_exp2f((s[v] - m) * 1.4426950408889634f);
            den += e;
            num = fmaf(e, s[v], num);
        }
        return num * __builtin_amdgcn_rcpf(den);   // den >= 1 (the max term contributes exp(0))
    }
}

// Two channels at once for the brick forward kernels (VALU-bound: scripts/loop_histogram.py).  Softmax with the exponentials taken
// relative to VIEW 0 instead of the maximum:
//   out = (s_0 + sum_{v>0} e_v s_v) / (1 + sum_{v>0} e_v),   e_v = exp(s_v - s_0)
// V - 1 v_exp_f32 instead of V and no v_max3 / v_max (3.5 ns and 2.0 ns per wave instruction against 1.1 for an fma:
// scripts/microbench_ops.hip).  Algebraically the same weights; what the max form buys is range: here e_v overflows once
// s_v - s_0 > 88.7.  Every way that can go wrong ends non-finite somewhere: an overflowed e_v makes den = Inf; finite e_v whose sum
// overflows make den = Inf; an overflowed e_v s_v makes num and the quotient Inf or NaN; non-finite samples make everything NaN.
// One class test on  den_a * den_b + (r_a + r_b)  therefore catches every such case of either channel (Inf - Inf reads NaN: still
// caught; den_a * den_b >= 2^128 with both quotients fine is a false alarm, only slow), and the wave redoes both channels in the
// max form -- a wave-uniform branch.  Underflow is harmless: e_v -> 0 is the limit of the weight.
// Issue order (MVHMR_AGG_GROUP, default 1): the transcendentals of both channels are issued BACK TO BACK -- 2 (V - 1) v_exp_f32,
// later the two v_rcp_f32 -- instead of wherever the scheduler drops them among the subtractions and fmas: on gfx950 a v_exp_f32
// between plain VALU instructions costs ~2 ns more than in a run of its kind (scripts/microbench_trans.hip: 4 exp + 12 fma per
// iteration take 31.3 ns grouped, 38.9 ns as exp, fma, fma, fma; the sum of the parts is 29.2).  Same operations, same results.
template <int METHOD, int V>
__device__ __forceinline__ void aggregate2(const float (&sa)[V], const float (&sb)[V], float &ra, float &rb)
{
    if constexpr (METHOD == AGG_SOFTMAX && V > 1 && MVHMR_AGG_GROUP) {
        float ta[V], tb[V];
#pragma unroll
        for (int v = 1; v < V; ++v) {
            ta[v] = (sa[v] - sa[0]) * 1.4426950408889634f;
            tb[v] = (sb[v] - sb[0]) * 1.4426950408889634f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 1; v < V; ++v) { ta[v] = __builtin_amdgcn_exp2f(ta[v]); tb[v] = __builtin_amdgcn_exp2f(tb[v]); }
        __builtin_amdgcn_sched_barrier(0);
        float da = 1.f, db = 1.f, na = sa[0], nb = sb[0];
#pragma unroll
        for (int v = 1; v < V; ++v) {
            da += ta[v]; na = fmaf(ta[v], sa[v], na);
            db += tb[v]; nb = fmaf(tb[v], sb[v], nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float ia = __builtin_amdgcn_rcpf(da), ib = __builtin_amdgcn_rcpf(db);
        __builtin_amdgcn_sched_barrier(0);
        ra = na * ia;
        rb = nb * ib;
        if (__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(fmaf(da, db, ra + rb)) < __builtin_inff())) != 0) {
            ra = aggregate<METHOD, V>(sa);
            rb = aggregate<METHOD, V>(sb);
        }
    } else if constexpr (METHOD == AGG_SOFTMAX && V > 1) {
        auto one = [](const float (&s)[V], float &den) __attribute__((always_inline)) {
            float num = s[0];
            den = 1.f;
#pragma unroll
            for (int v = 1; v < V; ++v) {
                const float e = __builtin_amdgcn_exp2f((s[v] - s[0]) * 1.4426950408889634f);
                den += e;
                num = fmaf(e, s[v], num);
            }
            return num * __builtin_amdgcn_rcpf(den);
        };
        float da, db;
        ra = one(sa, da);
        rb = one(sb, db);
        if (__builtin_amdgcn_ballot_w64(!(__builtin_fabsf(fmaf(da, db, ra + rb)) < __builtin_inff())) != 0) {
            ra = aggregate<METHOD, V>(sa);
            rb = aggregate<METHOD, V>(sb);
        }
    } else {
        ra = aggregate<METHOD, V>(sa);
        rb = aggregate<METHOD, V>(sb);
    }
}

// d(aggregate)/d(s_v) * g for every view, autograd of the same graph.
template <int METHOD, int V>
__device__ __forceinline__ void aggregate_grad(const float (&s)[V], float g, float (&ds)[V])
{
    if constexpr (METHOD == AGG_SUM) {
#pragma unroll
        for (int v = 0; v < V; ++v) ds[v] = g;
    } else if constexpr (METHOD == AGG_MEAN) {
        const float gv = __fdiv_rn(g, (float)V);
#pragma unroll
        for (int v = 0; v < V; ++v) ds[v] = gv;
    } else if constexpr (METHOD == AGG_MAX) {
        int am = 0;
#pragma unroll
        for (int v = 1; v < V; ++v) am = s[v] > s[am] ? v : am;   // first arg-max, as torch.max(dim)
#pragma unroll
        for (int v = 0; v < V; ++v) ds[v] = v == am ? g : 0.f;
    } else {
        // the same max as the forward's aggregate<> (no canonicalising v_max x,x in front of every operand, v_max3 where three
        // operands are at hand): the value is identical
        float m = s[0];
        if constexpr (V >= 3) {
            m = vmax3(s[0], s[1], s[2]);
#pragma unroll
            for (int v = 3; v + 1 < V; v += 2) m = vmax3(m, s[v], s[v + 1]);
            if constexpr ((V - 3) % 2 == 1) m = vmax(m, s[V - 1]);
        } else if constexpr (V == 2) {
            m = vmax(s[0], s[1]);
        }
        float e[V], den = 0.f, num = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            e[v] = __builtin_amdgcn_exp2f((s[v] - m) * 1.4426950408889634f);       // difference first, as in aggregate<>
            den += e[v];
            num = fmaf(e[v], s[v], num);
        }
        const float rden = __builtin_amdgcn_rcpf(den);
        const float o = num * rden, gr = g * rden;
        const float c0 = gr - gr * o;
#pragma unroll
        for (int v = 0; v < V; ++v) ds[v] = e[v] * fmaf(gr, s[v], c0);      // g * p_v * (1 + s_v - out)
    }
}

// Running form for a view count only known at run time (V > 8): one pass, same result up to rounding.
template <int METHOD>
struct RunningAgg {
    float m, den, num;
    __device__ __forceinline__ void init() { m = -INFINITY; den = 0.f; num = 0.f; }
    __device__ __forceinline__ void push(float s)
    {
        if constexpr (METHOD == AGG_SUM || METHOD == AGG_MEAN) num = __fadd_rn(num, s);
        else if constexpr (METHOD == AGG_MAX) m = fmaxf(m, s);
        else {
            const float mn = fmaxf(m, s);
            const float c = __expf(m - mn), e = __expf(s - mn);   // exp(-inf) = 0 on the first push
            den = fmaf(den, c, e);
            num = fmaf(num, c, e * s);
            m = mn;
        }
    }
    __device__ __forceinline__ float result(int V) const
    {
        if constexpr (METHOD == AGG_SUM) return num;
        else if constexpr (METHOD == AGG_MEAN) return __fdiv_rn(num, (float)V);
        else if constexpr (METHOD == AGG_MAX) return m;
        else return num * __builtin_amdgcn_rcpf(den);
    }
};

// ---------------------------------------------------------------- typed 4-channel vectors
struct alignas(16) f32x4 { float v[4]; };

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float *p)
    {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        return f32x4{{t.x, t.y, t.z, t.w}};
    }
    static __device__ __forceinline__ void store(float *p, const f32x4 &a)
    {
        *reinterpret_cast<float4 *>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    }
};
template <> struct Vec4<__half> {
    static __device__ __forceinline__ f32x4 load(const __half *p)
    {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        const __half2 lo = *reinterpret_cast<const __half2 *>(&raw.x), hi = *reinterpret_cast<const __half2 *>(&raw.y);
        const float2 a = __half22float2(lo), b = __half22float2(hi);
        return f32x4{{a.x, a.y, b.x, b.y}};
    }
    static __device__ __forceinline__ void store(__half *p, const f32x4 &a)
    {
        __half2 lo = __floats2half2_rn(a.v[0], a.v[1]), hi = __floats2half2_rn(a.v[2], a.v[3]);
        uint2 raw;
        raw.x = *reinterpret_cast<unsigned *>(&lo);
        raw.y = *reinterpret_cast<unsigned *>(&hi);
        *reinterpret_cast<uint2 *>(p) = raw;
    }
};

// bf16 is a storage type of the VOLUME only (out / grad_out with fp32 features): clang's native __bf16, whose conversion from float is
// v_cvt_pk_bf16_f32 on gfx950 (round to nearest even, NaN stays NaN)
typedef __bf16 bf16_t;

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<__half>(__half x) { return __half2float(x); }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
// fp16 results are the fp32 result rounded to nearest even, whatever produced it: the empty asm keeps hipcc from folding the last fp32
// multiply into the conversion (v_fma_mixlo_f16 rounds the exact product once, which differs from "fp32, then fp16" exactly at ties)
template <> __device__ __forceinline__ __half from_f32<__half>(float x) { asm("" : "+v"(x)); return __float2half_rn(x); }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// two floats -> one dword of a 16-bit storage type (low half = a)
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<__half>(float a, float b)
{
    asm("" : "+v"(a), "+v"(b));                                                  // see from_f32<__half>
    const __half2 h = __floats2half2_rn(a, b);                                   // round to nearest even
    return __builtin_bit_cast(unsigned, h);
}
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float a, float b)
{
    typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 h = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, h);
}

__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uniform(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

}  // namespace mvhmr
