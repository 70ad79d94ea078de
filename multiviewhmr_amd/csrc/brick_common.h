// Helpers shared by the brick kernels (forward: brick_fwd_kernel.h / brick_fwd_groups.h, backward: unproject_brick_bwd.hip).
#pragma once
#include <type_traits>
#include "device_common.h"

namespace mvhmr {

constexpr int kBZ = 32;            // z extent of a brick: 128-B output runs
constexpr int kBX = 4;
constexpr int kMaxChunks = 5;      // 64-slot DMA chunks per wave per quad (1024-thread bricks)
// 512-thread bricks (8 views) have half the waves for up to twice the window slots
constexpr int brick_chunks_per_wave(int nt) { return nt >= 1024 ? kMaxChunks : 2 * kMaxChunks; }
// View counts between the compiled ones run the next larger kernel with the missing views ABSENT (brick kernels: 2, 4, 8 view slots):
// an absent view has no camera and no window; where the aggregate needs it out of the way (softmax, max) its samples read
// kAbsentSample from slot kAbsentSlot of the always-zero head of every LDS ring buffer (its neighbours stay zero).
constexpr int brick_view_slots(int v) { return v <= 2 ? 2 : v <= 4 ? 4 : 8; }

// the four tiles of a sample's bricks (block order of k_fwd_ws, and of k_bwd_brick: see there): which two axes are halved and the tile's extents in bricks
struct BrickTiles { int hx, hy, hz, share; bool split_x, split_y, split_z; };
__host__ __device__ inline BrickTiles brick_tiles(int nbx, int nby, int nbz)
{
    BrickTiles t;
    t.split_z = nbz >= 2;
    t.split_x = t.split_z ? nbx > nby : true;
    t.split_y = t.split_z ? !t.split_x : true;
    t.hx = t.split_x ? (nbx + 1) / 2 : nbx; t.hy = t.split_y ? (nby + 1) / 2 : nby; t.hz = t.split_z ? (nbz + 1) / 2 : nbz;
    t.share = t.hx * t.hy * t.hz;
    return t;
}

constexpr int kAbsentSlot = 64;
constexpr float kAbsentSample = -3.4028234663852886e38f;                        // -FLT_MAX: exp(that - m) = 0, 0 * that = -0, never a maximum
constexpr int kZeroSlots = 128;    // always-zero 16-B slots at the head of every ring buffer (row stride <= 126)
constexpr int kZeroBytes = kZeroSlots * 16;

// workgroup barrier that waits for this wave's LDS operations only (not for global loads / stores in flight)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// workgroup barrier that waits for nothing: LDS reads of OTHER buffers may stay in flight across it
__device__ __forceinline__ void bare_barrier()
{
    asm volatile("s_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) void lds_void_t;

// f(integral_constant<int, I>) for I = FROM .. TO - 1, unrolled at compile time (indices that must reach immediate operands)
template <int FROM, int TO, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (FROM < TO) {
        f(std::integral_constant<int, FROM>{});
        static_for<FROM + 1, TO>(f);
    }
}

// LDS-DMA of 16 B per lane: lane l's bytes land at lds_dst + 16*l (lds_dst wave-uniform), read from base + voff[l].
// Written as inline asm on purpose: hipcc tracks the builtin form as a pending LDS write and puts s_waitcnt vmcnt(0)
// in front of every later ds_read, which drains the whole ring and the output stores each quad.  The asm form is
// outside its bookkeeping; completion is counted by hand (wait_vmcnt) before the barrier that precedes the reads.
__device__ __forceinline__ void glds16(const void *base, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_dst)
                 : "memory");
}

// same, m0 left clobbered (declared): no save / restore around every piece
__device__ __forceinline__ void glds16_m0(const void *base, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}

__device__ __forceinline__ f32x4 lds_tap(const unsigned char *smem, int addr)
{
    const float4 t = *reinterpret_cast<const float4 *>(smem + addr);
    return f32x4{{t.x, t.y, t.z, t.w}};
}

// 4 x 4 transpose across an aligned lane quad: on entry lane j holds r[i] = value(channel i, z_j); on exit lane j holds
// r[i] = value(channel j, z_i) -- 4 consecutive z of ONE channel, i.e. 16 contiguous bytes of the output.
// Two butterfly stages over DPP quad_perm (lane ^ 1, lane ^ 2): 4 moves + 12 selects, no LDS.
__device__ __forceinline__ void quad_transpose(float (&r)[4], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    auto xchg = [](float x, int ctrl) {
        return __builtin_bit_cast(float, ctrl == 1 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false)
                                                   : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
    };
    // stage A (partner = lane ^ 1): even lanes collect channels 0 / 2, odd lanes channels 1 / 3, for z_{j} and z_{j^1}
    {
        const float ya = xchg(b0 ? r[0] : r[1], 1), yb = xchg(b0 ? r[2] : r[3], 1);
        const float a0 = b0 ? ya : r[0], a1 = b0 ? r[1] : ya, a2 = b0 ? yb : r[2], a3 = b0 ? r[3] : yb;
        r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
    }
    // stage B (partner = lane ^ 2): lanes 0,1 keep the first channel of their pair and fetch its z_2, z_3; lanes 2,3 the second
    {
        const float ya = xchg(b1 ? r[0] : r[2], 2), yb = xchg(b1 ? r[1] : r[3], 2);
        const float c0 = b1 ? ya : r[0], c1 = b1 ? yb : r[1], c2 = b1 ? r[2] : ya, c3 = b1 ? r[3] : yb;
        r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in 0..20 (anything larger waits for everything): wait until at most n vector-memory operations of this wave
// (loads, LDS-DMA and stores count together, in issue order) are still outstanding
__device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (uniform(n)) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// no-return integer add in LDS (ds_add_u32: ~4-6 cycles per wave instruction on gfx950; ds_add_f32: ~190)
__device__ __forceinline__ void lds_add(int *p, int v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ds_add_u32 (no return) at an absolute LDS byte address + immediate offset.  Outside hipcc's lgkmcnt bookkeeping: completion is
// the next s_waitcnt lgkmcnt(0) (lds_barrier)
template <int OFF>
__device__ __forceinline__ void lds_add_at(unsigned addr, int v)
{
    asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}

// floor(x + 0.5) as int32 in one instruction
__device__ __forceinline__ int round_int(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int VT>
struct BrickShared {
    int bbox[VT][4];               // xmin, ymin, xmax, ymax of the nw taps (valid voxels only)
    float proj[VT][12];
    int aux[13];                   // backward: block-wide max |ds| per channel (float bits), one set of 4 per quad in flight (2 or 3); [12] tap multiplicity
};

__device__ __forceinline__ int wave_min(int x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int y = __shfl_xor(x, m); x = y < x ? y : x; }
    return x;
}
__device__ __forceinline__ int wave_max(int x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int y = __shfl_xor(x, m); x = y > x ? y : x; }
    return x;
}

// max over the wave without LDS traffic (the __shfl_xor form goes through ds_bpermute): four DPP steps leave every lane
// of a 16-lane row with the row's max, four readlanes and scalar max finish.  Result wave-uniform.
__device__ __forceinline__ int wave_max_dpp(int x)
{
    auto step = [](int v, auto ctrl) {
        const int y = __builtin_amdgcn_update_dpp(v, v, decltype(ctrl)::value, 0xF, 0xF, false);
        return y > v ? y : v;
    };
    x = step(x, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    x = step(x, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    x = step(x, std::integral_constant<int, 0x141>{});     // row_half_mirror
    x = step(x, std::integral_constant<int, 0x140>{});     // row_mirror
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const int c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// max over the wave, left in the lanes of the last 16-lane row (lane 63 among them): six fused v_max_i32_dpp -- four inside the rows
// (quad_perm, mirrors), row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3.  DPP moves and integer max both issue at half the
// fma rate on gfx950 (scripts/microbench_ops.hip), so the fused form halves the cost of the mov + max pairs hipcc makes of
// update_dpp; s_nop 1 = the two wait states a DPP read needs after a VALU write of its source.  x >= 0.
__device__ __forceinline__ int wave_max_to_last_row(int x)
{
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(x));
    return x;
}

// ds_max_i32 of lane 63's value into one LDS word (byte address, wave-uniform).  Written with an explicit exec mask: behind
// `if (lane == 63) atomicMax(...)` hipcc's atomic optimiser builds its own wave reduction (v_mbcnt, v_readfirstlane, compares).
// Completion: the next s_waitcnt lgkmcnt(0) (lds_barrier).
__device__ __forceinline__ void lds_max_from_lane63(unsigned lds_addr, int v)
{
    unsigned long long save;
    unsigned a;
    asm volatile("v_mov_b32 %1, %3\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "ds_max_i32 %1, %2\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save), "=&v"(a)
                 : "v"(v), "s"(lds_addr), "s"(0x8000000000000000ull)
                 : "memory");
}

}  // namespace mvhmr
