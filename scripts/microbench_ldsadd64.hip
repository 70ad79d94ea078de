// Round 4: ds_add_u64 (two channels per instruction, 32-bit fields with a sign-extended low field) against two ds_add_u32 on the
// backward's add pattern: a wave = 4 voxel columns x 16 z, z-neighbours 1.45 words apart, columns in rows `rs` words apart, and a share
// of instructions in which two columns fall on the SAME image column (same words for the z where the rows coincide).
// 1024 threads per CU like the kernel.  build: hipcc --offload-arch=gfx950 -O3 scripts/microbench_ldsadd64.hip -o mb64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WIDE>
__global__ void __launch_bounds__(1024) k(const int *lane_word, float *out, int iters)
{
    __shared__ long long acc[8192];                                              // 64 KB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) long long *)acc;
    if (WIDE) {
        const unsigned a = base + (unsigned)((lane_word[lane] + wave * 512) & 8191) * 8u;      // one 64-bit word per pixel (two channels)
        const long long v = ((long long)(lane + 1) << 32) + (long long)(-(lane + 3));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("ds_add_u64 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(u * 16) : "memory");   // 8 x 2 channels
        }
    } else {
        const unsigned a = base + (unsigned)((lane_word[lane] + wave * 512) & 8191) * 4u;      // planar: one 32-bit word per pixel and channel
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a), "v"(lane + 1), "n"((u & 1) * 32768 + (u >> 1) * 8) : "memory");   // 16 x 1 channel
        }
    }
    __syncthreads();
    long long s = 0;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s += acc[i];
    if (s == 12345) out[threadIdx.x] = (float)s;
}

template <int WIDE>
int run(const char *name, const int *h)
{
    int *d; float *out;
    CK(hipMalloc(&d, 64 * 4)); CK(hipMalloc(&out, 4096 * 4));
    CK(hipMemcpy(d, h, 64 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    hipLaunchKernelGGL(k<WIDE>, dim3(256), dim3(1024), 0, 0, d, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<WIDE>, dim3(256), dim3(1024), 0, 0, d, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // both forms move 16 channel-contributions per lane and iteration
    printf("%-28s %-52s %.3f ms -> %.2f ns per 16 channel-adds of a wave, per CU\n", WIDE ? "8 x ds_add_u64 (2 channels)" : "16 x ds_add_u32", name, ms,
           ms * 1e6 / ((double)iters * 16));
    CK(hipFree(d)); CK(hipFree(out));
    return 0;
}

int main()
{
    int w[64];
    auto z32 = [](int l5) { return l5 < 4 ? l5 : l5 < 12 ? 12 + l5 : l5 < 16 ? l5 - 8 : l5 < 20 ? 8 + l5 : l5 < 28 ? l5 - 12 : l5; };
    for (int l = 0; l < 64; ++l) w[l] = l;
    if (run<0>("consecutive words", w) || run<1>("consecutive words", w)) return 1;
    for (int rs : {27, 45}) {
        char name[96];
        for (int l = 0; l < 64; ++l) { const int zz = z32(l & 31), col = (l >> 5) * 2 + (zz >> 4), z = zz & 15; w[l] = col * rs + (int)floor(1.45 * z + 0.3 * col); }
        snprintf(name, 96, "kernel map, 4 columns in rows %d words apart", rs);
        if (run<0>(name, w) || run<1>(name, w)) return 1;
        // two of the four voxel columns on the same image column, rows offset by 0.5 px: about half of their z share a word
        for (int l = 0; l < 64; ++l) { const int zz = z32(l & 31), col = (l >> 5) * 2 + (zz >> 4), z = zz & 15, c2 = col == 3 ? 2 : col; w[l] = c2 * rs + (int)floor(1.45 * z + (col == 3 ? 0.5 : 0.3 * col)); }
        snprintf(name, 96, "same, columns 2 and 3 on one image column");
        if (run<0>(name, w) || run<1>(name, w)) return 1;
    }
    return 0;
}
