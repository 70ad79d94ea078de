// Does v_pk_fma_f32 double FP32 FMA throughput on MI355X?  (cycles per wave-instruction per SIMD at 4 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float *out, int iters, float seed)
{
    float2v a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i].x = seed + i + threadIdx.x * 1e-3f; a[i].y = a[i].x * 0.5f; }
    float2v m = {1.0001f, 0.9999f}, c = {0.0003f, 0.0001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(m), "v"(c)); }
            if (OP == 1) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i].x) : "v"(a[i].x), "v"(m.x), "v"(c.x)); }
            if (OP == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(m)); }
            if (OP == 3) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c)); }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int OP> void run(const char *name)
{
    float *out; hipMalloc(&out, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256, threads = 1024;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * 4;
    printf("%-14s %.3f ms -> %.2f cycles@2.4GHz per wave-instr per SIMD\n", name, ms, ms * 1e6 / instr_per_simd * 2.4);
}
int main() { run<1>("v_fma_f32"); run<0>("v_pk_fma_f32"); run<2>("v_pk_mul_f32"); run<3>("v_pk_add_f32"); return 0; }
