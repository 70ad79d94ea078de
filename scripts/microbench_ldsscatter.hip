// ds_add_u32 / ds_add_u64 under the backward kernel's scatter pattern on MI355X: a wave = NCOL z columns x (64 / NCOL) lanes; a lane's
// word = column offset + row(z) * stride, row(z) = floor(1.3 z) (the ~1.3 px per voxel of the north-star geometry), stride odd.
//   MODE 0: four ds_add_u32 into four planes (the shipped scheme, one channel per instruction)
//   MODE 1: two ds_add_u64 into two planes of 64-bit words (two channels per instruction)
//   SPREAD 0: the wave's columns hit the same pixels (neighbouring voxel columns); 1: unrelated offsets (columns 4 voxels apart)
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_ldsscatter.hip -o scripts/microbench_ldsscatter
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kPlane = 3392;                 // words per plane (the kernel's kZeroSlots + cap)

template <int NCOL, int MODE, int SPREAD>
__global__ void __launch_bounds__(1024) k(int *out, int iters)
{
    extern __shared__ int planes[];          // 4 x kPlane words (MODE 1: 2 x kPlane 64-bit words = the same bytes)
    for (int i = threadIdx.x; i < 4 * kPlane; i += blockDim.x) planes[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LZ = 64 / NCOL;
    const int c = lane / LZ, z = lane % LZ;
    const int row = (int)(1.3f * z);
    const int offs[4] = {0, 517, 1130, 1777};
    const int coff = SPREAD ? offs[c % 4] + 7 * c : c / 2;           // SPREAD 0: the same pixels (every second column one pixel further)
    int w = (wave * 131 + coff + row * 21) % (kPlane - 64);
    const int v = 1 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                    // the four taps: +0, +1, +stride, +stride+1
            const int a = w + (t & 1) + (t >> 1) * 21;
            if (MODE == 0) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) __hip_atomic_fetch_add(&planes[ch * kPlane + a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                unsigned long long *p64 = reinterpret_cast<unsigned long long *>(planes);
#pragma unroll
                for (int cp = 0; cp < 2; ++cp)
                    __hip_atomic_fetch_add(&p64[cp * kPlane + a], ((unsigned long long)v << 32) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        w = (w + 3) % (kPlane - 64);
    }
    __syncthreads();
    int s = 0;
    for (int i = threadIdx.x; i < 4 * kPlane; i += blockDim.x) s += planes[i];
    if (s == 12345) out[threadIdx.x] = s;
}

template <int NCOL, int MODE, int SPREAD>
int run(const char *name)
{
    int *out; CK(hipMalloc(&out, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256, threads = 1024;
    const size_t lds = 4 * kPlane * 4;
    hipLaunchKernelGGL((k<NCOL, MODE, SPREAD>), dim3(blocks), dim3(threads), lds, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NCOL, MODE, SPREAD>), dim3(blocks), dim3(threads), lds, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double chan_adds_per_cu = (double)iters * 16 * (threads / 64);      // wave-wide (tap, channel) adds
    printf("%-60s %.3f ms -> %.2f cycles per wave-wide (tap, channel) add per CU @2.4GHz\n", name, ms, ms * 1e6 / chan_adds_per_cu * 2.4);
    CK(hipFree(out));
    return 0;
}

int main()
{
    run<2, 0, 0>("u32, 2 columns x 32 z, same pixels");
    run<2, 0, 1>("u32, 2 columns x 32 z, unrelated");
    run<4, 0, 0>("u32, 4 columns x 16 z, same pixels");
    run<4, 0, 1>("u32, 4 columns x 16 z, unrelated (shipped)");
    run<1, 0, 1>("u32, 1 column x 64 z");
    run<2, 1, 0>("u64 (2 channels), 2 columns x 32 z, same pixels");
    run<2, 1, 1>("u64 (2 channels), 2 columns x 32 z, unrelated");
    run<4, 1, 0>("u64 (2 channels), 4 columns x 16 z, same pixels");
    run<4, 1, 1>("u64 (2 channels), 4 columns x 16 z, unrelated");
    run<1, 1, 1>("u64 (2 channels), 1 column x 64 z");
    return 0;
}
