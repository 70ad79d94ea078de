// Fused un-projection forward, WAVE-SPECIALISED brick kernel (gfx950 / CDNA4 only) -- round 5.
//
// Reference semantics: models/aggregation.py:20-87 (projection, depth mask, bilinear grid_sample with zero padding, cross-view
// aggregate); the arithmetic order of the projection and of the bilinear sample is pinned in device_common.h.
//
// Why.  profiles/r05_fwd_ablations.txt: with the arithmetic removed, the LDS-DMA window fills and the volume stores of k_fwd_brick
// take 2.8-2.9 ms of its 3.3 ms by themselves (stores alone 1.7 = the HBM write rate, window fills alone 1.1, and the two do not
// overlap: both live off the ~88 requests a CU keeps in flight), the arithmetic alone 2.5 ms -- and the two ADD UP to 3.3 because
// every wave does both: a wave whose store or LDS-DMA instruction waits for room in the vector-memory FIFOs issues no arithmetic.
// A timing build in which 4 of the 16 waves issued every memory instruction and the other 12 only computed ran at the memory floor.
//
// Mapping.
//   block    = one 8 x 8 x 32 brick, 1 024 threads = 2 MEMORY waves + 14 COMPUTE waves (<= 128 VGPRs; three layouts were built: the kernel's comment);
//   compute  = a wave owns 2 or 3 "units" (two x-adjacent voxel columns x 32 z: lane = (x parity, z)) of the brick's 32.  Tap records
//              (two LDS addresses + four weights per voxel and view) live in registers for all C / 4 channel quads.  A job
//              (quad, voxel) is 16 ds_read_b128 + 64 FMAs + the aggregate; its four results go to the result buffer R in LDS as
//              four ds_write_b32.  Compute waves never issue a vector-memory instruction;
//   memory   = per quad: (A) wait for its LDS-DMA pieces of this quad's windows, barrier A (publishes the windows), read last quad's
//              results from R (32 x ds_read_b128 over the memory waves), raise the "R is free" counter, request the next quad's windows
//              (global_load_lds_dwordx4, pieces wave, wave + NMW, ...), store last quad's results (32 x buffer_store_dwordx4 over the memory waves:
//              one instruction = 8 rows y x 128 B of one channel plane).  These waves spend their time waiting for FIFO room --
//              that is their job;
//   LDS      = two window buffers EXACTLY 64 512 B apart (the quad loop is unrolled by two and the second buffer is addressed
//              through the 16-bit offset field of ds_read_b128: no per-quad address arithmetic), R = 32 KiB behind them;
//   windows  = as k_fwd_brick: per view the bounding box of the brick's taps, column-major, rows split by parity.
// Softmax (PRE = true): the layout pass has multiplied the staged features by log2(e), so e_v = exp2(t_v - t_0) costs one
// subtraction, and ln 2 is folded into the final multiply; the overflow test is one compare per job (sum of the four denominators
// below 2^60: then no e_v * t_v can overflow either) with the max form as the wave-uniform fallback.  <= 2e-7 relative to the
// unscaled form; sum / mean / max read an unscaled copy and stay bit-exact.
#pragma once
#include "brick_fwd_groups.h"

namespace mvhmr {

constexpr int kWsMemWaves = 2, kWsComputeWaves = 14;                 // the wave layout of k_fwd_ws (see the kernel)
constexpr int kWsBufBytes = 64512;                                   // window buffer: zero region + 3 904 slots; < 65 536: a DS offset
constexpr int kWsCapSlots = (kWsBufBytes - kZeroBytes) / 16;
constexpr int kWsResBytes = 4 * 64 * 32 * 4;                         // R[channel][column = x * 8 + y][z] fp32
constexpr int kWsLdsBytes = 2 * kWsBufBytes + kWsResBytes;           // FwdShared<VT> behind it
constexpr int kWsSyncOff = kWsLdsBytes + 1024;                       // one LDS word behind FwdShared: the memory waves' "R has been read" counter
constexpr int kWsChunkTotal = 64;                                    // LDS-DMA pieces per quad (64 x 64 slots >= cap), dealt to the memory waves
static_assert(kWsCapSlots % 64 == 0 && kWsCapSlots <= kWsChunkTotal * 64, "window pool / chunk table");

// window geometry of a brick (block-uniform; the same arithmetic in both roles and in k_brick_gate)
template <int VT>
struct WsWindows {
    int wx0[VT], wy0[VT], ws[VT], whp[VT], slot0[VT], nch[VT + 1];
    int whr[VT], wbw[VT];        // half-rows / columns that taps can reach (whp is rounded up to 8, the last chunk to 64 slots: the rest is padding)
    bool fits;
};

template <int VT>
__device__ __forceinline__ void ws_size_windows(const FwdShared<VT> *sh, WsWindows<VT> &w)
{
    int used = 0, max_stride = 0;
    auto size = [&](bool round8) __attribute__((always_inline)) {
        w.nch[0] = 0; used = 0; max_stride = 0;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
            const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
            const int y0w = ymin & ~1;                                           // origin row even: rows split by parity
            int bw = 0, hp = 0;
            if (xmax >= xmin) { bw = xmax - xmin + 2; hp = (ymax + 3 - y0w) >> 1; }   // taps reach x0 + 1, y0 + 1
            w.whr[v] = hp; w.wbw[v] = bw;
            if (round8) hp = (hp + 7) & ~7;
            const int stride = 2 * hp, chunks = (stride * bw + 63) >> 6;
            w.wx0[v] = xmin; w.wy0[v] = y0w; w.ws[v] = stride; w.whp[v] = hp;
            max_stride = stride > max_stride ? stride : max_stride;
            w.slot0[v] = used;
            used += chunks << 6;
            w.nch[v + 1] = w.nch[v] + chunks;
        }
    };
    size(true);                                                                  // hp = 0 mod 8: conflict-free tap reads (sim_lds5b.py)
    if (!(used <= kWsCapSlots && max_stride + 2 <= kZeroSlots)) size(false);
    w.fits = used <= kWsCapSlots && max_stride + 2 <= kZeroSlots;
}

// two dwords at an LDS byte address + two immediate offsets in units of 256 B (outside hipcc's lgkmcnt bookkeeping: see write_half)
template <int O0, int O1>
__device__ __forceinline__ void lds_write2_at(unsigned addr, float v0, float v1)
{
    asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(v0), "v"(v1), "n"(O0), "n"(O1) : "memory");
}

// what both roles of a block know about their brick
template <typename TO>
struct WsBrick {
    const float4 *fk;            // staged features of this sample, quad 0
    TO *obase;                   // output of this sample, channel 0
    long long N;
    int b, kx, ky, kz, nq, nqv, C, H, W, X, Y, Z, nv;   // nq: whole channel quads (the fast loop's); nqv = (C + 3) / 4: the staged copy's quads per view
    unsigned chan_bytes, lds_base;
};

// the cold path of a compute wave: its units' voxels one by one through ws_slow_voxel
template <int METHOD, int VT, typename TO>
__device__ __attribute__((noinline)) void ws_slow_units(const WsBrick<TO> B, const float (*proj)[12], const Coords coords, int u0, int nunits, int dcol,
                                                        int zin, float unscale)
{
    const int vz = B.kz * kBZ + zin;
    for (int u = 0; u < nunits; ++u) {
        const int unit = u0 + u;
        const int vx = B.kx * 8 + 2 * (unit & 3) + dcol, vy = B.ky * 8 + (unit >> 2);
        if (vx < B.X && vy < B.Y && vz < B.Z)
            fwd_global_voxel<METHOD, VT, TO>(B.fk, B.obase, proj, coords, B.b, B.N, (unsigned)(((long long)vx * B.Y + vy) * B.Z + vz), 0, B.nq, B.nqv, B.C,
                                             B.H, B.W, B.nv, unscale);
    }
}

// ==================================================================== memory waves (NMW of them; wave = 0 .. NMW - 1)
template <int VT, typename TO, int NMW>
__device__ __forceinline__ void ws_memory_role(unsigned char *smem, const FwdShared<VT> *sh, const WsBrick<TO> &B, int wave, int lane)
{
    constexpr int MC = kWsChunkTotal / NMW;                                      // LDS-DMA pieces per wave and quad
    WsWindows<VT> win;
    ws_size_windows<VT>(sh, win);
    if (!win.fits) return;                                                       // the compute waves sample from global memory
    const int HW = B.H * B.W;
    // ---- LDS-DMA pieces of this wave: piece c = wave + NMW rr covers 64 consecutive slots of one view's window (the views are
    // packed back to back in 64-slot chunks, so piece c lands at slot 64 c of the buffer)
    unsigned go[MC];
    int n_m = 0;
#pragma unroll
    for (int rr = 0; rr < MC; ++rr) {
        const int c = wave + rr * NMW;
        go[rr] = 0;
        if (c < win.nch[VT]) {
            int v = 0;
#pragma unroll
            for (int uu = 1; uu < VT; ++uu) v += c >= win.nch[uu] ? 1 : 0;
            int sv = win.ws[0], ox = win.wx0[0], oy = win.wy0[0], c0 = win.nch[0], hv = win.whp[0], hr = win.whr[0], bwv = win.wbw[0];
#pragma unroll
            for (int uu = 1; uu < VT; ++uu)
                if (v == uu) { sv = win.ws[uu]; ox = win.wx0[uu]; oy = win.wy0[uu]; c0 = win.nch[uu]; hv = win.whp[uu]; hr = win.whr[uu]; bwv = win.wbw[uu]; }
            const int slot = ((c - c0) << 6) + lane;
            const int px = slot / sv;
            int py = slot - px * sv;
            // padding -- half-rows past the ones taps can reach (whp is rounded up to 8 for conflict-free tap reads), columns past the window
            // in the last 64-slot chunk --: never read, so never fetched (bit 0 of the offset: the lane sits out of the LDS-DMA)
            const bool dead = px >= bwv || (py >= hv ? py - hv : py) >= hr;
            py = py >= hv ? 2 * (py - hv) + 1 : 2 * py;                          // slot inside the column -> row (even rows first)
            int gx = ox + px, gy = oy + py;                                      // pad rows / columns past the window / outside the
            gx = gx < 0 ? 0 : (gx > B.W - 1 ? B.W - 1 : gx);                     // image: clamp -- those slots only meet zero weights
            gy = gy < 0 ? 0 : (gy > B.H - 1 ? B.H - 1 : gy);
            go[rr] = (unsigned)((v * B.nqv) * HW + gx * B.H + gy) * 16u | (dead ? 1u : 0u);
            ++n_m;
        }
    }
    const float4 *src_n = B.fk;                                                  // plane of the next quad to request
    auto dma = [&](int boff) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < MC; ++rr)
            if (rr < n_m) {
                const unsigned dst = B.lds_base + (unsigned)(boff + kZeroBytes + (wave + rr * NMW) * 1024);
                if (!(go[rr] & 1u)) glds16_m0(src_n, go[rr], dst);                // (lane-divergent: an exec mask around the piece)
            }
        src_n += HW;
    };
    // ---- stores.  fp32 volume: instruction j = wave + NMW k (k < NS = 32 / NMW) writes channel j >> 3, brick row y = j & 7: lane = (x, z quad)
    // reads R[j >> 3][(j & 7) * 8 + x][4 z4 ..] (1 KiB contiguous per instruction) and writes 16 B; 8 lanes = one 128-B run of z.
    // 16-bit volume: instruction j (k < NS = 16 / NMW) writes channel j >> 2, brick rows y = 2 (j & 3) + (lane >> 5): lane = (y bit, x, z octet)
    // reads 8 results (32 B of R), rounds them once (from_f32's rounding) and writes 16 B; 4 lanes = one 64-B run of z.
    constexpr bool kWide = sizeof(TO) == 4;
    constexpr int NS = (kWide ? 32 : 16) / NMW, NR = kWide ? NS : 2 * NS;         // store instructions / 16-B reads of R per wave and quad
    const int xl = kWide ? lane >> 3 : (lane >> 2) & 7, yl = kWide ? 0 : lane >> 5, zl = kWide ? (lane & 7) * 4 : (lane & 3) * 8;
    const int vx = B.kx * 8 + xl, vz = B.kz * kBZ + zl;
    const unsigned voff = (vx < B.X && vz < B.Z) ? (unsigned)((((long long)vx * B.Y + B.ky * 8 + yl) * B.Z + vz) * (long long)sizeof(TO)) : kDropOffset;   // beyond num_records: dropped
    const unsigned ystep = (unsigned)(B.Z * (int)sizeof(TO));
    int n_st = 0;                                                                // store instructions of this wave per quad
#pragma unroll
    for (int k = 0; k < NS; ++k) n_st += B.ky * 8 + (kWide ? (wave + NMW * k) & 7 : 2 * ((wave + NMW * k) & 3)) < B.Y ? 1 : 0;
    const int r_rd = 2 * kWsBufBytes + (kWide ? lane * 16 : yl * 1024 + xl * 128 + zl * 4);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    dma(0);                                                                      // quad 0 -> buffer 0
    for (int q = 0; q <= B.nq; ++q) {
        // the pieces of quad q have landed: younger than them are only the n_st stores issued behind them (none before quad 2)
        if (q < B.nq) {
            if (q < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else wait_vmcnt_ladder<0, NS, NS>(n_st);
        }
        bare_barrier();                                                          // A(q): windows of quad q published; results of quad q - 1 complete
        // (the only barrier of the quad loop: the hand-back "R may be overwritten" is a counter, see below)
        float4 res[NR];
        if (q > 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int jj = wave + NMW * k;
                if constexpr (kWide) {
                    res[k] = *reinterpret_cast<const float4 *>(smem + r_rd + jj * 1024);
                } else {
                    const int at = r_rd + (jj >> 2) * 8192 + (jj & 3) * 2048;
                    res[2 * k] = *reinterpret_cast<const float4 *>(smem + at);
                    res[2 * k + 1] = *reinterpret_cast<const float4 *>(smem + at + 16);
                }
            }
        }
        if (q < B.nq) {
            // R has been read (lgkmcnt(0)): tell the compute waves, which may then write quad q's results -- every lane adds 1, so the
            // counter stands at 64 NMW q once all memory waves have read the results of quad q - 1.  No barrier: nobody waits here.
            if (q > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_u32 %0, %1" : : "v"(kWsSyncOff), "v"(1) : "memory");
            if (q + 1 < B.nq) dma(((q + 1) & 1) * kWsBufBytes);                  // every wave has passed A(q): the other buffer is free
        }
        if (q > 0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(B.obase + (long long)((q - 1) * 4) * B.N, 0, (int)(4u * B.chan_bytes), 0x00020000);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int jj = wave + NMW * k;
                if constexpr (kWide) {
                    const int ys = jj & 7, ch = jj >> 3;
                    if (B.ky * 8 + ys < B.Y) {
                        const u32x4 d = {__builtin_bit_cast(unsigned, res[k].x), __builtin_bit_cast(unsigned, res[k].y),
                                         __builtin_bit_cast(unsigned, res[k].z), __builtin_bit_cast(unsigned, res[k].w)};
                        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)voff, (int)(ys * ystep + ch * B.chan_bytes), kStAux);
                    }
                } else {
                    const int ys = 2 * (jj & 3), ch = jj >> 2;
                    if (B.ky * 8 + ys < B.Y) {                                   // the pair's second row may still be outside: its lanes are dropped
                        const u32x4 d = {pack2<TO>(res[2 * k].x, res[2 * k].y), pack2<TO>(res[2 * k].z, res[2 * k].w),
                                         pack2<TO>(res[2 * k + 1].x, res[2 * k + 1].y), pack2<TO>(res[2 * k + 1].z, res[2 * k + 1].w)};
                        const unsigned vo = B.ky * 8 + ys + yl < B.Y ? voff : kDropOffset;
                        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)vo, (int)(ys * ystep + ch * B.chan_bytes), kStAux);
                    }
                }
            }
        }
    }
}

// ==================================================================== compute waves: NVOX "units" (two x-adjacent columns x 32 z) each,
// units u0 .. u0 + NVOX - 1 of the brick's 32 (unit = 4 y + x pair).  Ends with the barrier sequence the memory waves run.
template <int METHOD, int VT, typename TO, bool PRE, int NVOX>
__device__ __forceinline__ void ws_compute_role(unsigned char *smem, FwdShared<VT> *sh, const WsBrick<TO> &B, const Coords &coords, int u0, int lane,
                                                int ctid, int cthreads)
{
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;
    constexpr bool kRel = METHOD == AGG_SOFTMAX && VT > 1;                       // views >= 1 are folded as differences to view 0 (ws_softmax_pair)
    const int nv = B.nv;
    int dcol, zin;
    fwd_lane_voxel<1>(lane, dcol, zin);                                          // lane = 32 * (x parity) + z (z quads permuted: LDS pass groups are z runs)
    const int vz_r = B.kz * kBZ + zin;
    const int vz = vz_r < B.Z ? vz_r : B.Z - 1;
    unsigned vox[NVOX];
    bool inside[NVOX];
    float w00[NVOX][VT], w01[NVOX][VT], w10[NVOX][VT], w11[NVOX][VT];
    int txy[NVOX][VT];
    unsigned valid = 0;
    {
        int bxmin[VT], bymin[VT], bxmax[VT], bymax[VT];
        const int big = 1 << 30;
#pragma unroll
        for (int v = 0; v < VT; ++v) { bxmin[v] = big; bymin[v] = big; bxmax[v] = -big; bymax[v] = -big; }
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            const int unit = u0 + u;
            const int vx_r = B.kx * 8 + 2 * (unit & 3) + dcol, vy_r = B.ky * 8 + (unit >> 2);
            inside[u] = vx_r < B.X && vy_r < B.Y && vz_r < B.Z;
            const int vx = vx_r < B.X ? vx_r : B.X - 1, vy = vy_r < B.Y ? vy_r : B.Y - 1;   // outside the volume: the clamped edge voxel's centre, no part in the windows
            vox[u] = (unsigned)(((long long)vx * B.Y + vy) * B.Z + vz);         // N < 2^28 (brick_fwd_supported)
            float c0, c1, c2;
            voxel_xyz(coords, B.b, B.N, vox[u], c0, c1, c2);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const Taps t = make_taps(sh->proj[v], c0, c1, c2, B.H, B.W);
                w00[u][v] = t.w00; w01[u][v] = t.w01; w10[u][v] = t.w10; w11[u][v] = t.w11;
                txy[u][v] = (t.rx0 & 0xFFFF) | (t.ry0 << 16);                  // nw tap, each in [-1, 32 766]: one register until the windows are known
                if (t.any && inside[u] && v < nv) {
                    valid |= 1u << (u * VT + v);
                    bxmin[v] = t.rx0 < bxmin[v] ? t.rx0 : bxmin[v]; bxmax[v] = t.rx0 > bxmax[v] ? t.rx0 : bxmax[v];
                    bymin[v] = t.ry0 < bymin[v] ? t.ry0 : bymin[v]; bymax[v] = t.ry0 > bymax[v] ? t.ry0 : bymax[v];
                }
                // one projection at a time: interleaved by the scheduler, the 4 NVOX independent make_taps spill ~100 B per lane
                // (0.4 GB of scratch writes per launch at the north star: profiles/r05_fwd_ablations.txt G)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int nxmin = wave_max_dpp(-bxmin[v]), nymin = wave_max_dpp(-bymin[v]);
            const int xmax = wave_max_dpp(bxmax[v]), ymax = wave_max_dpp(bymax[v]);
            if (lane == 0 && xmax >= -nxmin) {
                atomicMin(&sh->bbox[v][0], -nxmin); atomicMin(&sh->bbox[v][1], -nymin);
                atomicMax(&sh->bbox[v][2], xmax); atomicMax(&sh->bbox[v][3], ymax);
            }
        }
    }
    __syncthreads();
    WsWindows<VT> win;
    ws_size_windows<VT>(sh, win);

    if (!win.fits) {
        // ---- windows do not fit the LDS pool: sample straight from global memory (the memory waves have left).  ONE call that rebuilds
        // the voxel indices itself and is followed by the return: nothing of the fast path lives across it (the first form, a loop of calls
        // over vox[] / inside[], made the allocator spill ~50 registers per lane in front of this branch on EVERY brick: 0.4 GB of scratch
        // writes per launch at the north star)
        ws_slow_units<METHOD, VT, TO>(B, sh->proj, coords, u0, NVOX, dcol, zin, PRE ? kLn2 : 1.f);
        return;
    }

    // zero regions at the head of both window buffers (samples that are identically zero read them); absent views: kAbsentSample
    if (ctid == 0) *reinterpret_cast<int *>(smem + kWsSyncOff) = 0;
    for (int i = ctid; i < kZeroSlots * 2; i += cthreads) {
        const float z = (kAbsentReads && nv < VT && i % kZeroSlots == kAbsentSlot) ? kAbsentSample : 0.f;
        *reinterpret_cast<float4 *>(smem + (i / kZeroSlots) * kWsBufBytes + (i % kZeroSlots) * 16) = make_float4(z, z, z, z);
    }
    // ---- LDS byte offsets (inside a buffer) of the taps in column x0: a0 = the EVEN row of the footprint, a1 = the odd row (both in
    // one register: buffer-relative offsets are below 2^16); column x0 + 1 is one column stride further.  The weights are kept in that
    // order (even row x0, even row x0 + 1, odd row x0, odd row x0 + 1): for an odd y0 the sum runs sw, se, nw, ne instead of ATen's
    // nw, ne, sw, se -- <= 1 ulp of the sample.
    unsigned ap[NVOX][VT];
    unsigned aq[NVOX == 2 ? NVOX : 1][VT];                                        // two-unit waves have the registers: the odd-row address on its own (no unpacking)
    constexpr bool kUnpacked = NVOX == 2;
    int ws16[VT];
    // the column strides: in VGPRs where registers allow (v_add_u32 v, v, v issues at the full rate, v, s, v at half); the three-unit
    // waves of the 1 024-thread layout have none to spare and keep them scalar
    constexpr bool kStrideVgpr = NVOX != 3;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        int s16 = win.ws[v] * 16;
        if (kAbsentReads && v >= nv) s16 = 16;                                   // "one column further": the zero slot next to the absent sample
        if constexpr (kStrideVgpr) asm volatile("v_mov_b32 %0, %1" : "=v"(ws16[v]) : "s"(s16));
        else ws16[v] = s16;
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            const bool ok = (valid >> (u * VT + v)) & 1u;
            const int tx = (int)(short)(txy[u][v] & 0xFFFF), ty = txy[u][v] >> 16;
            const int yr = ty - win.wy0[v];
            const int sc = win.slot0[v] + (tx - win.wx0[v]) * win.ws[v];
            int a0 = ok ? kZeroBytes + (sc + ((yr + 1) >> 1)) * 16 : 0;
            int a1 = ok ? kZeroBytes + (sc + win.whp[v] + (yr >> 1)) * 16 : 0;
            if (yr & 1) {
                const float t0 = w00[u][v], t1 = w01[u][v];
                w00[u][v] = w10[u][v]; w01[u][v] = w11[u][v]; w10[u][v] = t0; w11[u][v] = t1;
            }
            if (kAbsentReads && v >= nv) {                                       // wave-uniform: the absent view's one "tap"
                a0 = kAbsentSlot * 16; a1 = kAbsentSlot * 16;
                w00[u][v] = kRel ? 0.5f : 1.f; w01[u][v] = 0.f; w10[u][v] = 0.f; w11[u][v] = 0.f;   // (relative softmax: -FLT_MAX / 2 - s0 stays finite, so 0 * d is -0, not NaN)
            }
            if constexpr (kUnpacked) { ap[u][v] = (unsigned)a0; aq[u][v] = (unsigned)a1; }
            else ap[u][v] = (unsigned)a0 | ((unsigned)a1 << 16);
        }
    }
    // R address of this lane's unit u0, channel 0: R[channel][column = 2 unit + x parity][z]; unit u0 + u is 256 u bytes further,
    // channel i 64 columns further: immediate offsets
    const int rbase = 2 * kWsBufBytes + ((2 * u0 + dcol) * 32 + zin) * 4;

    f32x4 T[2][4];
    float sq[4][VT], sp[4][VT], res[4];
    auto read_view = [&](auto boff, int u, int v, int set) __attribute__((always_inline)) {
        constexpr int BOFF = decltype(boff)::value;
        asm volatile("" : "+v"(ap[u][v]));                                       // unpacked per use: hoisted out of the quad loop, the addresses would spill
        const unsigned pk = ap[u][v];
        int base, base1;
        if constexpr (kUnpacked) {
            asm volatile("" : "+v"(aq[u][v]));
            base = (int)pk; base1 = (int)aq[u][v];
        } else {
            base = (int)(pk & 0xFFFFu); base1 = (int)(pk >> 16);
        }
        const int far = base + ws16[v], far1 = base1 + ws16[v];
        T[set][0] = lds_tap(smem, base + BOFF); T[set][2] = lds_tap(smem, base1 + BOFF);
        T[set][1] = lds_tap(smem, far + BOFF); T[set][3] = lds_tap(smem, far1 + BOFF);
    };
    // the aggregate of one job in two halves (channels 0 / 1, then 2 / 3)
    auto agg_half = [&](float (&s)[4][VT], auto half) __attribute__((always_inline)) {
        constexpr int h = decltype(half)::value;
        if constexpr (METHOD == AGG_SOFTMAX && VT > 1) {
            float d;
            ws_softmax_pair<VT, PRE>(s[2 * h], s[2 * h + 1], res[2 * h], res[2 * h + 1], d);
            if (__builtin_amdgcn_ballot_w64(!(d < 1.152921504606847e18f)) != 0) {  // wave-uniform: redo the pair in the max form
                res[2 * h] = ws_softmax_safe<VT, PRE>(s[2 * h]);
                res[2 * h + 1] = ws_softmax_safe<VT, PRE>(s[2 * h + 1]);
            }
        } else if constexpr (METHOD == AGG_MEAN) {
            res[2 * h] = __fdiv_rn(aggregate<AGG_SUM, VT>(s[2 * h]), (float)nv);
            res[2 * h + 1] = __fdiv_rn(aggregate<AGG_SUM, VT>(s[2 * h + 1]), (float)nv);
        } else {
            res[2 * h] = aggregate<METHOD, VT>(s[2 * h]);
            res[2 * h + 1] = aggregate<METHOD, VT>(s[2 * h + 1]);
        }
    };
    // two of a job's four results.  ds_write2st64_b32 with the (channel, unit) parts of the two addresses in the offset fields, written as inline asm:
    // left to the compiler the twelve addresses become twelve loop-invariant registers.  Outside its lgkmcnt bookkeeping, which is
    // harmless (LDS operations retire in order: an unknown younger write only makes a counted wait cover more); lds_barrier() ends the quad.
    auto write_half = [&](auto utag, auto htag) __attribute__((always_inline)) {
        constexpr int u = decltype(utag)::value, h = decltype(htag)::value;
        // both channels of the pair in one instruction: ds_write2st64_b32, offsets in units of 256 B (a channel plane of R is 8 192 B = 32 units):
        // two LDS instructions per job instead of four, -1 % (profiles/r05_fwd_variants.txt)
        lds_write2_at<(2 * h) * 32 + u, (2 * h + 1) * 32 + u>((unsigned)rbase, res[2 * h], res[2 * h + 1]);
    };

    // One quad.  Jobs u = 0 .. NVOX - 1: request views 0 and 1, first half of the PREVIOUS job's aggregate (two results to R), fold view 0 /
    // request view 2, second half of the previous aggregate, fold view 1 / request view 3, fold views 2 and 3.  Barrier B sits in front of
    // the quad's first write to R.  The last job is aggregated behind the loop, then lgkmcnt(0) + barrier A of the next quad.
    // before a quad's first write to R: the memory waves must have read the previous quad's results (counter >= 64 NMW q).  The counter
    // is requested at the head of the job and looked at here, behind sixteen tap reads: normally it has long been raised
    // (ds_read_b32 as inline asm: through a volatile pointer hipcc makes it a FLAT load, whose wait drains every tap read in flight.
    // LDS operations return in order, so with at most 15 younger ones outstanding -- lgkmcnt(15) -- the counter has arrived)
    int r_need = 0;                                                              // 64 NMW q
    auto read_r_counter = [&]() __attribute__((always_inline)) {
        int seen;
        asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(kWsSyncOff) : "memory");
        return seen;
    };
    auto wait_r_free = [&](int seen) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(seen) : : "memory");
        while (uniform(seen) < r_need) {
            __builtin_amdgcn_s_sleep(1);
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(kWsSyncOff) : "memory");
        }
    };
    auto quad_iter = [&](auto boff) __attribute__((always_inline)) {
        auto job = [&](auto utag) __attribute__((always_inline)) {
            constexpr int u = decltype(utag)::value;
            auto &cur = (u & 1) ? sp : sq;
            auto &prev = (u & 1) ? sq : sp;
            int seen = 0;
            if constexpr (u == 1) seen = read_r_counter();
            read_view(boff, u, 0, 0);
            if constexpr (VT > 1) read_view(boff, u, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u > 0) {
                agg_half(prev, std::integral_constant<int, 0>{});
                if constexpr (u == 1) wait_r_free(seen);                         // the memory waves have read the previous quad's results
                write_half(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{}, std::integral_constant<int, 0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                if constexpr (u > 0) {
                    if (v == (VT > 1 ? 1 : 0)) {
                        agg_half(prev, std::integral_constant<int, 1>{});
                        write_half(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{}, std::integral_constant<int, 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (kRel && v > 0) {
                        cur[i][v] = bilerp_rel(T[v & 1][0].v[i], T[v & 1][1].v[i], T[v & 1][2].v[i], T[v & 1][3].v[i], w00[u][v], w01[u][v], w10[u][v], w11[u][v],
                                               cur[i][0]);
                    } else {
                        cur[i][v] = bilerp(T[v & 1][0].v[i], T[v & 1][1].v[i], T[v & 1][2].v[i], T[v & 1][3].v[i], w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    }
                    asm volatile("" : "+v"(cur[i][v]));                           // fold HERE: keeps the tap registers short-lived
                }
                __builtin_amdgcn_sched_barrier(0);
                if (v + 2 < VT) read_view(boff, u, v + 2, v & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        job(std::integral_constant<int, 0>{});
        if constexpr (NVOX > 1) job(std::integral_constant<int, 1>{});
        if constexpr (NVOX > 2) job(std::integral_constant<int, 2>{});
        if constexpr (NVOX > 3) job(std::integral_constant<int, 3>{});
        static_assert(NVOX <= 4, "jobs per quad");
        // the quad's last job (its samples are in sq for an odd NVOX, in sp for an even one)
        auto &last = ((NVOX - 1) & 1) ? sp : sq;
        agg_half(last, std::integral_constant<int, 0>{});
        if constexpr (NVOX == 1) wait_r_free(read_r_counter());
        write_half(std::integral_constant<int, NVOX - 1>{}, std::integral_constant<int, 0>{});
        agg_half(last, std::integral_constant<int, 1>{});
        write_half(std::integral_constant<int, NVOX - 1>{}, std::integral_constant<int, 1>{});
        r_need += 64 * kWsMemWaves;
        lds_barrier();                                                           // results written; A of the next quad
    };
    lds_barrier();                                                               // zero regions written; A(0)
    for (int q = 0; q < B.nq; q += 2) {
        quad_iter(std::integral_constant<int, 0>{});
        if (q + 1 < B.nq) quad_iter(std::integral_constant<int, kWsBufBytes>{});
    }
}

// Wave layout: 1 024 threads = 2 memory waves (waves 0, 1: SIMDs 0, 1) + 14 compute waves; waves w, w + 4, w + 8, w + 12 share a SIMD and every
// SIMD computes 8 units: SIMDs 0 / 1 three compute waves of 3 / 3 / 2 units, SIMDs 2 / 3 four of 2.  Three layouts were built and timed on one
// box (scripts/exp/patches/ws_wave_layouts.patch brings the other two back):
//   768 threads  = 4 memory waves + 8 compute waves of 4 units (<= 168 VGPRs)                                              3.48 ms
//   1 024 threads = 4 memory waves + 8 compute waves of 3 units + 4 of 2 (every SIMD: 1 memory wave + 3 / 3 / 2 units)     3.21 ms
//   1 024 threads = 2 memory waves + 14 compute waves (this one)                                                           3.18 ms
// (half of the three-unit waves -- the ones whose column strides have to stay in SGPRs, which halves the rate of the two v_add_u32 per voxel,
// view and quad that use them -- became two-unit waves; two memory waves keep up with 32 LDS-DMA pieces + 16 stores each)
template <int METHOD, int VT, typename TO, bool PRE>
__global__ void __launch_bounds__(1024)
k_fwd_ws(const float4 *__restrict__ featK, const float *__restrict__ proj, const Coords coords, TO *__restrict__ out, int C, int H, int W,
         int X, int Y, int Z, int nby, int nbz, int bricks_per_sample, int total_blocks, int nv, Gate gate)
{
    // nv <= VT views are real (3 views run the 4-view kernel): the others have no camera, no window and no part in the aggregate --
    // their samples read kAbsentSample from a slot of the zero region (softmax, max) or plain zeros (sum, mean)
    if (gated_off(gate)) return;
    static_assert(!PRE || METHOD == AGG_SOFTMAX, "only the softmax reads a prescaled copy");
    constexpr int NMW = kWsMemWaves, NCW = kWsComputeWaves;
    extern __shared__ __align__(16) unsigned char smem[];
    FwdShared<VT> *sh = reinterpret_cast<FwdShared<VT> *>(smem + kWsLdsBytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order (speed only): blocks i, i + 8, ... share an XCD under round-robin dispatch.  Four tiles per sample -- half of z (when
    // there are two bricks along z) x half of the longer of x / y, else half of x x half of y --, XCDs 0-3 on sample 2 p, XCDs 4-7 on sample
    // 2 p + 1: the 32 bricks an XCD works on at a time then share most of their windows, and a feature plane is fetched by at most four L2s.
    // Memory-side reads of the north star: 2.8 GB instead of the 4.5 GB of eight tiles per sample over all eight XCDs (1.3 GB are compulsory;
    // profiles/r05_fwd_ablations.txt G) -- at the same kernel time: those re-reads hit the Infinity Cache and cost little energy.
    const int nbx = bricks_per_sample / (nby * nbz);
    const BrickTiles T = brick_tiles(nbx, nby, nbz);
    const int bid = (int)blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int pj = j / T.share, r = j % T.share, tile = xcd & 3;
    int b = 2 * pj + (xcd >> 2);
    if ((2 * pj + 1) * bricks_per_sample >= total_blocks) {                      // an odd batch's last sample: all eight XCDs on it, XCD t and t + 4
        b = 2 * pj;                                                              // take alternate bricks of tile t
        if ((r & 1) != (xcd >> 2)) return;
    }
    const int cz = r % T.hz, cy = (r / T.hz) % T.hy, cx = r / (T.hz * T.hy);
    const int kz = T.split_z ? (tile >> 1) * T.hz + cz : cz;
    const int kx = T.split_x ? (T.split_z ? tile & 1 : tile >> 1) * T.hx + cx : cx;
    const int ky = T.split_y ? (tile & 1) * T.hy + cy : cy;
    if (kx >= nbx || ky >= nby || kz >= nbz || b * bricks_per_sample >= total_blocks) return;
    WsBrick<TO> B;
    B.N = (long long)X * Y * Z;
    B.nq = C >> 2; B.nqv = (C + 3) >> 2; B.C = C; B.H = H; B.W = W; B.X = X; B.Y = Y; B.Z = Z; B.nv = nv; B.b = b; B.kx = kx; B.ky = ky; B.kz = kz;
    B.obase = out + (long long)b * C * B.N;
    B.fk = featK + (long long)b * nv * B.nqv * (H * W);
    B.chan_bytes = (unsigned)(B.N * sizeof(TO));
    B.lds_base = (unsigned)(size_t)(lds_void_t *)smem;

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = tid < nv * 12 ? proj[((long long)b * nv) * 12 + tid] : 0.f;
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    __syncthreads();

    if (wave < NMW) {
        __syncthreads();                                                         // the compute waves' boxes are complete
        ws_memory_role<VT, TO, NMW>(smem, sh, B, wave, lane);
        return;
    }
    const int ctid = tid - NMW * 64;
    const int simd = wave & 3, idx = wave >> 2;
    if (simd < 2) {
        if (idx < 3) ws_compute_role<METHOD, VT, TO, PRE, 3>(smem, sh, B, coords, simd * 8 + (idx - 1) * 3, lane, ctid, NCW * 64);
        else ws_compute_role<METHOD, VT, TO, PRE, 2>(smem, sh, B, coords, simd * 8 + 6, lane, ctid, NCW * 64);
    } else {
        ws_compute_role<METHOD, VT, TO, PRE, 2>(smem, sh, B, coords, simd * 8 + idx * 2, lane, ctid, NCW * 64);
    }
}

// ---- host side
inline bool brick_fwd_ws_shape_impl(const Problem &p)
{
    // 3 or 4 views, z rows of whole 16-B store units (4 fp32 / 8 halves), enough bricks to fill the chip (fewer: k_fwd_brick splits the channels)
    if (brick_view_slots(p.V) != 4 || (p.Z & ((p.out_f16 || p.out_bf16) ? 7 : 3)) || p.X <= kBX) return false;
    const long long bricks = (long long)((p.X + 7) / 8) * ((p.Y + 7) / 8) * ((p.Z + kBZ - 1) / kBZ) * p.B;
    return bricks >= 256;
}

template <int METHOD, bool PRE, typename TO>
hipError_t launch_fwd_ws_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s)
{
    constexpr int VT = 4;
    const int nbx = (p.X + 7) / 8, nby = (p.Y + 7) / 8, nbz = (p.Z + kBZ - 1) / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    static_assert(sizeof(FwdShared<VT>) <= 1024 && kWsSyncOff + 16 <= 160 * 1024, "LDS layout");
    const size_t lds = (size_t)kWsSyncOff + 16;
    auto kern = k_fwd_ws<METHOD, VT, TO, PRE>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int grid = 8 * brick_tiles(nbx, nby, nbz).share * ((p.B + 1) / 2);          // 8 XCDs x bricks of a tile x pairs of samples
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, total, p.V,
                       make_gate(p, true));
    return hipGetLastError();
}

}  // namespace mvhmr
