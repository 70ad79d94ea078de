// True VALU issue cost in SHADER cycles (s_memtime) and the clock the chip holds (s_memtime / s_memrealtime x 100 MHz) on MI355X,
// 16 waves per CU, for the instruction kinds of the forward kernel: independent v_fma_f32 chains, v_exp_f32, v_pk_fma_f32, DPP moves.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *stamps, int iters, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
    float m = 1.0001f + seed * 1e-6f, c = 0.0003f + seed * 1e-6f;
    asm volatile("" : "+v"(m), "+v"(c));
    unsigned hbits = 0x3c003800u + (threadIdx.x & 1);                             // two halves: 1.0, 0.5
    asm volatile("" : "+v"(hbits));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], m, c);                          // 3 VGPR sources
            if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (OP == 2) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.0003f);              // literal operands
            if (OP == 4) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, a[i]), __builtin_bit_cast(int, a[(i + 8) & 15]), 0x104, 0xF, 0x5, false));
            if (OP == 5) a[i] = a[i] * m;
            if (OP == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(hbits), "v"(m));
            if (OP == 7) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(hbits), "v"(m));
        }
        if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 x = {a[i], a[i + 1]}, mm = {m, m}, cc = {c, c};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(mm), "v"(cc));
                a[i] = x.x; a[i + 1] = x.y;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
int run(const char *name, int threads, int per_iter)
{
    float *out; unsigned long long *st, h[512];
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&st, 512 * 8));
    const int iters = 20000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, st, 100, 1.f);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, st, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, st, 512 * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    const double waves_per_simd = threads / 256.0;
    const double instr = (double)iters * per_iter * waves_per_simd;
    printf("%-28s %4d thr: %.3f ms, clock %.2f GHz, %.2f shader cycles per wave-instr per SIMD\n", name, threads, ms, cyc / real * 0.1, cyc / 256 / instr);
    return 0;
}

int main()
{
    for (int t : {256, 512, 1024}) {
        run<0>("v_fma_f32 (3 VGPRs)", t, 16);
        run<2>("v_fma_f32 (literals)", t, 16);
        run<5>("v_mul_f32", t, 16);
        run<1>("v_exp_f32", t, 16);
        run<3>("v_pk_fma_f32", t, 8);
        run<4>("v_mov_b32_dpp row_shl:4 (bank mask)", t, 16);
        run<6>("v_fma_mix_f32 (lo half)", t, 16);
        run<7>("v_fma_mix_f32 (hi half)", t, 16);
    }
    return 0;
}
