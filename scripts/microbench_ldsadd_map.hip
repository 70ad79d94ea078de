// Round 4: which lanes does the LDS serve together in a ds_add_u32, i.e. which lane maps make the backward's adds conflict-free?
// A wave's 64 lanes = 4 voxel columns x 16 z; z-neighbours are 1.45 words apart along a plane row, the four columns sit in rows that are
// `rstride` words apart.  Lane maps: NAT lane = 16 column + z | Z32 the backward's renumbering (every ds_read_b128 pass group
// {0-3,12-15,20-27}, {4-11,16-19,28-31} holds 16 consecutive z) | probes with 8 / 16 / 32 consecutive lanes conflict-free by
// construction and the others colliding, to find the group size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k(const int *lane_word, float *out, int iters)
{
    __shared__ int acc[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = lane_word[lane] + wave * 1024;                               // 0 <= lane_word < 512
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) __hip_atomic_fetch_add(&acc[(w + u * 512) & 16383], lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    int s = 0;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += acc[i];
    if (s == 12345) out[threadIdx.x] = (float)s;
}

int run(const char *name, const int *h)
{
    int *d; float *out;
    CK(hipMalloc(&d, 64 * 4)); CK(hipMalloc(&out, 4096 * 4));
    CK(hipMemcpy(d, h, 64 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, threads = 1024;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %.3f ms -> %.2f ns per wave instruction per CU\n", name, ms, ms * 1e6 / ((double)iters * 8 * 16));
    CK(hipFree(d)); CK(hipFree(out));
    return 0;
}

int main()
{
    int w[64];
    auto z32 = [](int l5) { return l5 < 4 ? l5 : l5 < 12 ? 12 + l5 : l5 < 16 ? l5 - 8 : l5 < 20 ? 8 + l5 : l5 < 28 ? l5 - 12 : l5; };
    for (int l = 0; l < 64; ++l) w[l] = l;
    run("consecutive words (conflict-free)", w);
    for (int rs : {27, 29, 43, 45, 59}) {
        char name[128];
        for (int l = 0; l < 64; ++l) { const int col = l >> 4, z = l & 15; w[l] = col * rs + (int)floor(1.45 * z + 0.3 * col); }
        snprintf(name, 128, "NAT  lane = 16 col + z, rows %d words apart", rs); run(name, w);
        for (int l = 0; l < 64; ++l) { const int zz = z32(l & 31), col = (l >> 5) * 2 + (zz >> 4), z = zz & 15; w[l] = col * rs + (int)floor(1.45 * z + 0.3 * col); }
        snprintf(name, 128, "Z32  b128-group renumbering, rows %d words apart", rs); run(name, w);
    }
    // group-size probes: lanes inside a block of G consecutive lanes get distinct banks, lanes of different blocks the SAME bank set
    for (int G : {8, 16, 32, 64}) {
        char name[128];
        for (int l = 0; l < 64; ++l) w[l] = (l % G) + 64 * (l / G);              // block k: words 64k .. 64k+G-1 -> banks 0..G-1 again
        snprintf(name, 128, "probe: blocks of %d consecutive lanes share banks 0..%d", G, G - 1); run(name, w);
    }
    for (int G : {16, 32}) {                                                     // the same with the b128 pass groups instead of consecutive lanes
        char name[128];
        for (int l = 0; l < 64; ++l) { const int zz = z32(l & 31) + 32 * (l >> 5); w[l] = (zz % G) + 64 * (zz / G); }
        snprintf(name, 128, "probe: b128 pass groups (%d lanes) share banks 0..%d", G, G - 1); run(name, w);
    }
    return 0;
}
