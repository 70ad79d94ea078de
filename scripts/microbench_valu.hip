// VALU issue-rate calibration on MI355X: cycles per wave64 instruction per SIMD, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void k(float *out, int iters, float seed)
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
    const float m = 1.0001f, c = 0.0003f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], m, c);
            if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.5f;                // exp + mul
            if (OP == 2) a[i] = fmaxf(a[i] * m, c);                                 // mul + max
            if (OP == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + c;                    // rcp + add
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int OP>
int run(const char *name, int threads, int ops_per_iter)
{
    float *out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = threads / 64.0 / 4.0;
    const double instr_per_simd = (double)iters * 8 * ops_per_iter * waves_per_simd;
    printf("%-12s %4d thr/CU (%.0f waves/SIMD): %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, threads, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}

int main()
{
    for (int t : {256, 512, 1024}) {
        run<0>("fma", t, 1);
        run<1>("exp2+mul", t, 2);
        run<2>("mul+max", t, 2);
        run<3>("rcp+add", t, 2);
    }
    return 0;
}
