// Do LDS reads (ds_read_b128) and VALU work overlap on MI355X?  16 waves per CU, no barriers, no global traffic.
// Each iteration: R ds_read_b128 per lane (conflict-free or 2-way), F independent v_fma_f32; the loaded values are "used"
// (forcing the s_waitcnt) one iteration later, as a software-pipelined kernel would.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int F, int CONFLICT>
__global__ void __launch_bounds__(1024) k(float *out, int iters)
{
    __shared__ float4 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = (wave * 64 + (CONFLICT ? ((lane & 7) | ((lane >> 4) << 3)) * 2 + ((lane >> 3) & 1) * 16 : lane)) & 8191;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = lane * 1e-3f + i;
    float4 t[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = lds[(base + r * 1024) & 8191];
    for (int it = 0; it < iters; ++it) {
        float4 n[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) n[r] = lds[(base + r * 1024 + (it & 7) * 64) & 8191];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < F; ++f) a[f & 7] = __builtin_fmaf(a[f & 7], 1.0001f, 0.0003f);
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("" ::"v"(t[r].x), "v"(t[r].y), "v"(t[r].z), "v"(t[r].w));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = n[r];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}
template <int R, int F, int C> void run()
{
    float *out; hipMalloc(&out, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<R, F, C>), dim3(256), dim3(1024), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, F, C>), dim3(256), dim3(1024), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("R=%2d b128 reads, F=%3d fma per wave-iteration, %s: %.3f ms -> %.0f ns per iteration per CU (16 waves)\n", R, F, C ? "2-way conflicts" : "conflict-free", ms, ms * 1e6 / iters);
    hipFree(out);
}
int main()
{
    run<0, 48, 0>(); run<4, 0, 0>(); run<4, 48, 0>(); run<4, 0, 1>(); run<4, 48, 1>();
    run<0, 96, 0>(); run<4, 96, 0>(); run<8, 96, 0>(); run<8, 0, 0>(); run<8, 96, 1>(); run<8, 0, 1>();
    run<16, 0, 0>(); run<16, 190, 0>(); run<0, 190, 0>(); run<16, 190, 1>(); run<16, 0, 1>();
    return 0;
}
