// Round 4: what a v_exp_f32 costs when it is MIXED with plain VALU (the softmax of the forward kernel: 16 transcendentals among ~150
// plain instructions per job).  Patterns of 16 independent instructions per iteration, 1 / 2 / 4 waves per SIMD; event time per wave
// instruction and SIMD.  Expected if costs simply add: n_exp * t_exp + n_fma * t_fma.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// PAT: 0 = 16 fma | 1 = 16 exp | 2 = (exp, fma) x 8 | 3 = 8 exp, 8 fma | 4 = (exp, fma, fma, fma) x 4 | 5 = 4 exp, 12 fma
//      6 = (exp, fma x 7) x 2 | 7 = dependent pairs: exp then fma on its result, x 8 | 8 = (rcp, fma) x 8
template <int PAT>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed * 0.01f + i * 0.001f + threadIdx.x * 1e-5f;
    float m = 0.9999f + seed * 1e-7f, c = 0.0003f + seed * 1e-6f;
    asm volatile("" : "+v"(m), "+v"(c));
    auto fma = [&](int i) __attribute__((always_inline)) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c)); };
    auto ex = [&](int i) __attribute__((always_inline)) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); };
    auto rc = [&](int i) __attribute__((always_inline)) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])); };
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0) { _Pragma("unroll") for (int i = 0; i < 16; ++i) fma(i); }
        if (PAT == 1) { _Pragma("unroll") for (int i = 0; i < 16; ++i) ex(i); }
        if (PAT == 2) { _Pragma("unroll") for (int i = 0; i < 16; i += 2) { ex(i); fma(i + 1); } }
        if (PAT == 3) { _Pragma("unroll") for (int i = 0; i < 8; ++i) ex(i); _Pragma("unroll") for (int i = 8; i < 16; ++i) fma(i); }
        if (PAT == 4) { _Pragma("unroll") for (int i = 0; i < 16; i += 4) { ex(i); fma(i + 1); fma(i + 2); fma(i + 3); } }
        if (PAT == 5) { _Pragma("unroll") for (int i = 0; i < 4; ++i) ex(i); _Pragma("unroll") for (int i = 4; i < 16; ++i) fma(i); }
        if (PAT == 6) { _Pragma("unroll") for (int i = 0; i < 16; i += 8) { ex(i); _Pragma("unroll") for (int j = 1; j < 8; ++j) fma(i + j); } }
        if (PAT == 7) { _Pragma("unroll") for (int i = 0; i < 16; i += 2) { ex(i); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i + 1]) : "v"(a[i]), "v"(m)); } }
        if (PAT == 8) { _Pragma("unroll") for (int i = 0; i < 16; i += 2) { rc(i); fma(i + 1); } }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int PAT>
int run(const char *name, int threads, int n_exp)
{
    float *out;
    CK(hipMalloc(&out, 4096 * 4));
    const int iters = 20000;
    hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(threads), 0, 0, out, 100, 1.f);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_iter_ns = ms * 1e6 / iters / (threads / 256.0);             // per wave and iteration of 16 instructions, per SIMD
    printf("%-40s %4d thr: %.3f ms  %.2f ns per 16 instructions per wave and SIMD (adds to %.2f)\n", name, threads, ms, per_iter_ns,
           n_exp * 4.03 + (16 - n_exp) * 1.15);
    CK(hipFree(out));
    return 0;
}

int main()
{
    for (int t : {256, 512, 1024}) {
        run<0>("16 fma", t, 0);
        run<1>("16 exp", t, 16);
        run<2>("(exp, fma) x 8", t, 8);
        run<3>("8 exp, 8 fma", t, 8);
        run<4>("(exp, fma, fma, fma) x 4", t, 4);
        run<5>("4 exp, 12 fma", t, 4);
        run<6>("(exp, 7 fma) x 2", t, 2);
        run<7>("(exp, dependent fma) x 8", t, 8);
        run<8>("(rcp, fma) x 8", t, 8);
    }
    return 0;
}
