// LDS floating-point atomic adds on MI355X under the two fp32 denormal modes (the MODE register's FP_DENORM field), and ds_add_f64:
// ds_add_f32 costs ~190 cycles per wave instruction with denormals enabled (HIP's default) -- is the flush-to-zero mode the fast one?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: ds_add_f32, default mode   1: ds_add_f32 after FP_DENORM[fp32] = flush   2: ds_add_f64   3: ds_add_u32   4: ds_add_rtn_f32 (result used)
// PAT 0: conflict-free   1: spread   2: pairs share an address
template <int PAT, int MODE>
__global__ void k(float *out, int iters)
{
    __shared__ double acc64[8192];
    float *acc = reinterpret_cast<float *>(acc64);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    if (MODE == 1) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0" ::: "memory");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int w = PAT == 0 ? lane : PAT == 1 ? (17 * lane) & 1023 : lane >> 1;
    w += wave * 1024;
    const float v = 1.f + lane;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float *p = &acc[(w + u * 64) & 16383];
            if (MODE == 0 || MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) __hip_atomic_fetch_add(&acc64[(w + u * 64) & 8191], (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<int *>(p), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else keep += __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    float s = keep;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += acc[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int PAT, int MODE>
int run(const char *name, int threads)
{
    float *out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 1000, blocks = 256;
    hipLaunchKernelGGL((k<PAT, MODE>), dim3(blocks), dim3(threads), 0, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<PAT, MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)iters * 8 * (threads / 64);
    printf("%-44s %4d thr/CU: %.3f ms -> %.1f cycles per wave-instr per CU @2.4GHz\n", name, threads, ms, ms * 1e6 / instr_per_cu * 2.4);
    return 0;
}

int main()
{
    for (int t : {256, 1024}) {
        run<0, 0>("ds_add_f32 denormals on, conflict-free", t);
        run<0, 1>("ds_add_f32 fp32 denormals flushed, conflict-free", t);
        run<1, 1>("ds_add_f32 fp32 denormals flushed, spread", t);
        run<2, 1>("ds_add_f32 fp32 denormals flushed, pairs", t);
        run<0, 2>("ds_add_f64 conflict-free", t);
        run<2, 2>("ds_add_f64 pairs", t);
        run<0, 3>("ds_add_u32 conflict-free", t);
        run<0, 4>("ds_add_rtn_f32 conflict-free", t);
    }
    return 0;
}
