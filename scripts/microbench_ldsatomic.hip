// LDS float-add throughput on MI355X: ds_add_f32 (no return) under different address patterns, against a plain
// ds_read_b32 + v_add + ds_write_b32 read-modify-write that is only legal when the wave owns the addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// PAT 0: lane l -> word l (conflict-free)   1: word 17*l mod 4096 (spread)   2: word (l/2) (pairs share an address)
// PAT 3: word 32*l (all lanes one bank)     4: all lanes one address
template <int PAT, int MODE>
__global__ void k(float *out, int iters)
{
    __shared__ unsigned long long acc64[8192];
    float *acc = reinterpret_cast<float *>(acc64);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int w = PAT == 0 ? lane : PAT == 1 ? (17 * lane) & 1023 : PAT == 2 ? lane >> 1 : PAT == 3 ? (32 * lane) & 1023 : 0;
    w += wave * 1024;
    const float v = 1.f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float *p = &acc[(w + u * 64) & 16383];
            if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f32 (result unused)
            else if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<int *>(p), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32
            else if (MODE == 3) __hip_atomic_fetch_add(&acc64[(w + u * 64) & 8191], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u64
            else { *p = *p + v; }
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += acc[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int PAT, int MODE>
int run(const char *name, int threads)
{
    float *out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, blocks = 256;
    hipLaunchKernelGGL((k<PAT, MODE>), dim3(blocks), dim3(threads), 0, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<PAT, MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)iters * 8 * (threads / 64);
    printf("%-34s %4d thr/CU: %.3f ms -> %.1f cycles per wave-instr per CU @2.4GHz\n", name, threads, ms, ms * 1e6 / instr_per_cu * 2.4);
    return 0;
}

int main()
{
    for (int t : {256, 1024}) {
        run<0, 0>("ds_add_f32 conflict-free", t);
        run<1, 0>("ds_add_f32 spread (17*l)", t);
        run<2, 0>("ds_add_f32 pairs share address", t);
        run<3, 0>("ds_add_f32 one bank", t);
        run<4, 0>("ds_add_f32 one address", t);
        run<0, 2>("ds_add_u32 conflict-free", t);
        run<1, 2>("ds_add_u32 spread", t);
        run<2, 2>("ds_add_u32 pairs share address", t);
        run<3, 2>("ds_add_u32 one bank", t);
        run<4, 2>("ds_add_u32 one address", t);
        run<0, 3>("ds_add_u64 conflict-free", t);
        run<2, 3>("ds_add_u64 pairs share address", t);
        run<0, 1>("read+add+write conflict-free", t);
        run<1, 1>("read+add+write spread", t);
    }
    return 0;
}
