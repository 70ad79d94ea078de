// Global atomic-add throughput on MI355X by operand type: f32 vs u32 vs u64, one element per lane, each wave-instruction
// 64 consecutive elements of a pseudo-random row of a large table (the shape of a gradient-window flush).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T>
__global__ void k(T *tab, long long rows, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long h = wid * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < per_wave; ++i) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        const long long row = (long long)((h >> 20) % (unsigned long long)rows);
        T *p = tab + row * 64 + lane;
        if constexpr (sizeof(T) == 4 && !__is_same(T, unsigned)) __hip_atomic_fetch_add(p, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p, (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
int run(const char *name, size_t table_bytes)
{
    T *tab; CK(hipMalloc(&tab, table_bytes)); CK(hipMemset(tab, 0, table_bytes));
    const long long rows = table_bytes / (64 * sizeof(T));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, threads = 512, per_wave = 2000;
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, tab, rows, 20);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, tab, rows, per_wave);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)blocks * (threads / 64) * per_wave * 64;
    printf("%-6s table %5zu MB: %.3f ms -> %.1f G adds/s, %.2f TB/s of added bytes\n", name, table_bytes >> 20, ms, n / ms * 1e-6, n * sizeof(T) / ms * 1e-9);
    CK(hipFree(tab));
    return 0;
}

int main()
{
    for (size_t mb : {32, 1200}) {
        run<float>("f32", mb << 20);
        run<unsigned>("u32", mb << 20);
        run<unsigned long long>("u64", mb << 20);
    }
    return 0;
}
