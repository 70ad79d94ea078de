// Does an exec-masked ds_add_u32 cost less?  Same conflict-free pattern with 64 / 32 / 16 active lanes per wave instruction, 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KEEP>                 // lanes with (lane % KEEP) == 0 stay active; KEEP = 1: all
__global__ void __launch_bounds__(1024) k(int *out, int iters)
{
    __shared__ int acc[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = lane + wave * 1024;
    if (lane % KEEP == 0)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __hip_atomic_fetch_add(&acc[(w + u * 64) & 16383], lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    int s = 0;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += acc[i];
    if (s == 12345) out[threadIdx.x] = s;
}

template <int KEEP>
int run(const char *name)
{
    int *out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    hipLaunchKernelGGL(k<KEEP>, dim3(256), dim3(1024), 0, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KEEP>, dim3(256), dim3(1024), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %.3f ms -> %.2f ns per wave instruction per CU\n", name, ms, ms * 1e6 / ((double)iters * 8 * 16));
    return 0;
}

int main()
{
    run<1>("ds_add_u32, 64 lanes active");
    run<2>("ds_add_u32, 32 lanes active");
    run<4>("ds_add_u32, 16 lanes active");
    run<8>("ds_add_u32,  8 lanes active");
    return 0;
}
