// Microbenchmark: HBM write efficiency of the (B,C,X,Y,Z) output when a 256-thread block owns a voxel brick
// (bx,by,bz) and writes one float per thread per channel.  Answers: how long must the contiguous z-run be?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BX, int BY, int BZ, bool NT>
__global__ void __launch_bounds__(BX * BY * BZ) k_write(float *out, int S, int C, int bricks_per_sample)
{
    const int brick = blockIdx.x % bricks_per_sample, b = blockIdx.x / bricks_per_sample;
    const int nbz = S / BZ, nby = S / BY;
    const int kz = brick % nbz, ky = (brick / nbz) % nby, kx = brick / (nbz * nby);
    const int t = threadIdx.x;
    const int z = kz * BZ + t % BZ, y = ky * BY + (t / BZ) % BY, x = kx * BX + t / (BZ * BY);
    const long long N = (long long)S * S * S;
    const long long n = ((long long)x * S + y) * S + z;
    float *p = out + (long long)b * C * N + n;
    float v = (float)t;
    for (int c = 0; c < C; ++c) {
        if (NT) __builtin_nontemporal_store(v, p + (long long)c * N);
        else p[(long long)c * N] = v;
        v += 1.f;
    }
}

template <int BX, int BY, int BZ, bool NT>
int run(float *out, int B, int S, int C, const char *name)
{
    const int bps = (S / BX) * (S / BY) * (S / BZ);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_write<BX, BY, BZ, NT>), dim3(B * bps), dim3(BX * BY * BZ), 0, 0, out, S, C, bps);
    CK(hipEventRecord(e0));
    const int iters = 5;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_write<BX, BY, BZ, NT>), dim3(B * bps), dim3(BX * BY * BZ), 0, 0, out, S, C, bps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double bytes = (double)B * C * S * S * S * 4;
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    return 0;
}

int main()
{
    const int B = 32, S = 64, C = 256;
    float *out;
    CK(hipMalloc(&out, (size_t)B * C * S * S * S * 4));
    run<1, 4, 64, false>(out, B, S, C, "1x4x64 (256B runs)");
    run<2, 4, 32, false>(out, B, S, C, "2x4x32 (128B runs)");
    run<4, 4, 16, false>(out, B, S, C, "4x4x16 (64B runs)");
    run<4, 8, 8, false>(out, B, S, C, "4x8x8 (32B runs)");
    run<8, 8, 4, false>(out, B, S, C, "8x8x4 (16B runs)");
    run<1, 4, 64, true>(out, B, S, C, "1x4x64 nt");
    run<2, 4, 32, true>(out, B, S, C, "2x4x32 nt");
    run<4, 4, 16, true>(out, B, S, C, "4x4x16 nt");
    run<4, 8, 8, true>(out, B, S, C, "4x8x8 nt");
    run<8, 8, 8, false>(out, B, S, C, "8x8x8 512thr (32B runs)");
    run<4, 4, 32, false>(out, B, S, C, "4x4x32 512thr (128B runs)");
    run<4, 4, 64, false>(out, B, S, C, "4x4x64 1024thr (256B runs)");
    return 0;
}
