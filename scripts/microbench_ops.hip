// Issue cost (ns per wave instruction per SIMD, 16 waves per CU) of the instruction kinds the brick backward spends its VALU time on:
// conversions, integer address arithmetic, compares / selects, DPP reductions.  Independent chains over 16 registers, asm volatile
// so that the compiler adds nothing (ops that write vcc / an SGPR get an s_nop 0 from the hazard recogniser: marked Y).  v_fma_f32 is the yardstick (1.15 ns, scripts/microbench_clock.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OPS(X, Y) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
    X(1, "v_cvt_rpi_i32_f32", "v_cvt_rpi_i32_f32 %0, %0") \
    X(2, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0") \
    X(3, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0") \
    X(4, "v_rndne_f32", "v_rndne_f32 %0, %0") \
    X(5, "v_max_i32", "v_max_i32 %0, %0, %1") \
    X(6, "v_and_b32", "v_and_b32 %0, %0, %1") \
    X(7, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1") \
    X(8, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2") \
    X(9, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
    X(10, "v_add_u32", "v_add_u32 %0, %0, %1") \
    X(11, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2") \
    X(12, "v_rcp_f32", "v_rcp_f32 %0, %0") \
    X(13, "v_cndmask_b32 (s[22:23])", "v_cndmask_b32 %0, %0, %1, s[22:23]") \
    Y(14, "v_cmp_gt_f32 (s[20:21])", "v_cmp_gt_f32 s[20:21], %0, %1") \
    X(15, "v_max_i32 row_shr:1 (dpp)", "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(16, "v_mov_b32 (plain)", "v_mov_b32 %0, %1") \
    X(17, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
    X(18, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
    X(19, "v_sub_f32 + abs", "v_sub_f32 %0, |%0|, %1") \
    X(20, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0") \
    X(21, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 5") \
    X(22, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
    X(23, "v_exp_f32", "v_exp_f32 %0, %0") \
    X(24, "v_fmac_f32", "v_fmac_f32 %0, %1, %2") \
    X(25, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2") \
    X(26, "v_lshlrev_b32", "v_lshlrev_b32 %0, 2, %0") \
    X(27, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2") \
    X(28, "v_fma_f32 neg/abs mods", "v_fma_f32 %0, -%0, |%1|, %2") \
    Y(29, "v_readlane_b32 -> s", "v_readlane_b32 s20, %0, 3") \
    X(30, "v_fmamk_f32 (literal)", "v_fmamk_f32 %0, %0, 0x3fb8aa3b, %1") \
    X(31, "v_fmaak_f32 (literal)", "v_fmaak_f32 %0, %0, %1, 0x3fb8aa3b") \
    X(32, "v_max_f32", "v_max_f32 %0, %0, %1") \
    X(33, "v_min_f32", "v_min_f32 %0, %0, %1") \
    X(34, "v_or_b32", "v_or_b32 %0, %0, %1") \
    X(35, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
    X(36, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
    X(37, "v_mul_f32 (literal)", "v_mul_f32 %0, 0x3fb8aa3b, %0") \
    X(38, "v_mul_f32 (sgpr)", "v_mul_f32 %0, s24, %0") \
    X(39, "v_fma_f32 (sgpr)", "v_fma_f32 %0, %0, s24, %1") \
    X(41, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1") \
    X(42, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2") \
    X(43, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %0, %1") \
    X(44, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0") \
    Y(45, "v_add_co_u32 (vcc)", "v_add_co_u32 %0, vcc, %0, %1") \
    X(46, "v_mov_b32 quad_perm (dpp)", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(47, "v_add_f32 row_shr:1 (dpp)", "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(48, "v_exp_f16", "v_exp_f16 %0, %0") \
    X(49, "v_add_f32", "v_add_f32 %0, %0, %1") \
    X(50, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
    X(51, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %2") \
    Y(52, "v_cmp_gt_f32 + v_cndmask (vcc)", "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc") \
    X(53, "v_mac-like v_fmac_f32 (sgpr)", "v_fmac_f32 %0, s24, %1") \
    X(54, "v_sub_f32", "v_sub_f32 %0, %0, %1") \
    X(55, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0") \
    X(56, "v_max_u32", "v_max_u32 %0, %0, %1") \
    X(57, "v_lshrrev_b32", "v_lshrrev_b32 %0, 1, %0") \
    X(58, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 2")

template <int OP>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float seed)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
    float m = 1.0001f + seed * 1e-6f, c = 0.0003f + seed * 1e-6f;
    asm volatile("" : "+v"(m), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#define X(n, name, text) if (OP == n) asm volatile(text : "+v"(a[i]) : "v"(m), "v"(c));
#define Y(n, name, text) if (OP == n) asm volatile(text : "+v"(a[i]) : "v"(m), "v"(c) : "vcc", "s20", "s21");
            OPS(X, Y)
#undef X
#undef Y
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int OP>
int run(const char *name)
{
    float *out;
    CK(hipMalloc(&out, 4096 * 4));
    const int iters = 20000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, 100, 1.f);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-30s %.3f ms  %.2f ns per wave instruction per SIMD\n", name, ms, ms * 1e6 / ((double)iters * 16 * 4));
    CK(hipFree(out));
    return 0;
}

int main()
{
#define X(n, name, text) if (run<n>(name)) return 1;
    OPS(X, X)
#undef X
    return 0;
}
