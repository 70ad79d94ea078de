// Do narrower-scope global float atomics execute in the XCD's L2 on MI355X?  global_atomic_add_f32 at agent / workgroup / wavefront
// scope, 256 contiguous bytes per wave instruction, into a table that every XCD shares or that is split per XCD (blocks i, i+8, ...
// run on one XCD under round-robin dispatch, so a per-XCD slice is only ever touched from one L2).  Sums are verified: a scope that
// loses adds is visible.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SCOPE, bool SPLIT>
__global__ void k(float *tab, long long rows, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long h = wid * 0x9E3779B97F4A7C15ull + 12345;
    const long long rows_x = SPLIT ? rows / 8 : rows, row0 = SPLIT ? (blockIdx.x & 7) * rows_x : 0;
    for (int i = 0; i < per_wave; ++i) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const long long row = row0 + (long long)((h >> 20) % (unsigned long long)rows_x);
        float *p = tab + row * 64 + lane;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (SCOPE == 3) asm volatile("global_atomic_add_f32 %0, %1, off sc0" :: "v"(p), "v"(1.f) : "memory");
        if (SCOPE == 4) asm volatile("global_atomic_add_f32 %0, %1, off nt" :: "v"(p), "v"(1.f) : "memory");
    }
}

template <int SCOPE, bool SPLIT>
int run(const char *name, size_t table_bytes)
{
    float *tab; CK(hipMalloc(&tab, table_bytes)); CK(hipMemset(tab, 0, table_bytes));
    const long long rows = table_bytes / 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, threads = 512, per_wave = 1000;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SCOPE, SPLIT>), dim3(blocks), dim3(threads), 0, 0, tab, rows, per_wave);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)blocks * (threads / 64) * per_wave * 64;
    std::vector<float> h(table_bytes / 4);
    CK(hipMemcpy(h.data(), tab, table_bytes, hipMemcpyDeviceToHost));
    double sum = 0;
    for (float x : h) sum += x;
    printf("%-34s table %5zu MB %s: %.3f ms -> %.2f TB/s of added bytes; adds found %.6f of issued\n", name, table_bytes >> 20, SPLIT ? "split per XCD" : "shared       ", ms,
           n * 4 / ms * 1e-9, sum / n);
    CK(hipFree(tab));
    return 0;
}

int main()
{
    for (size_t mb : {16, 256, 1200}) {
        run<0, false>("agent scope", mb << 20);
        run<0, true>("agent scope", mb << 20);
        run<1, false>("workgroup scope", mb << 20);
        run<1, true>("workgroup scope", mb << 20);
        run<2, true>("wavefront scope", mb << 20);
        run<3, true>("asm sc0", mb << 20);
        run<4, true>("asm nt", mb << 20);
    }
    return 0;
}
