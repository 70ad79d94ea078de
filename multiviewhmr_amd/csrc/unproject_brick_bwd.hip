// "brick" variant of the backward pass (gradient w.r.t. the feature maps): the default wherever the brick forward runs.
//
// The gather backward scatters every tap with a global float atomic: 4 taps x 4 B per (voxel, view, channel) = 137 GB of
// atomic traffic at the north-star size against a chip-wide rate of ~1.3 TB/s -- ~0.1 s.  Here the same bricks and
// windows as the forward (brick_fwd_kernel.h) are used to sum on chip first:
//   per brick and channel quad   re-sample the brick's voxels from the LDS-staged feature window (recompute, nothing
//                                is saved by the forward), apply the aggregate's Jacobian (aggregate_grad), and add
//                                ds * w into a GRADIENT window in LDS (planar per channel: a wave's lanes -- consecutive
//                                z, i.e. consecutive slots of a column-major window line -- fall into consecutive
//                                banks).  LDS float atomics run at ~190 cycles
//                                per wave instruction on gfx950, integer ones at 4-6 (scripts/microbench_ldsatomic.hip):
//                                the window is accumulated in FIXED POINT, one power-of-two scale per channel and quad
//                                chosen from the block-wide max |ds| and the brick's tap multiplicity, ds_add_u32;
//   then                         flush the gradient window: one global float atomic per window pixel and channel, issued
//                                as 16 pixels x 4 channels = 256 contiguous bytes of the quad-planar accumulator per wave
//                                instruction (the full-rate shape on gfx950) -- ~20 GB instead of 137 GB.
// Windows, staged copy and accumulator are COLUMN-major quad-planar (B,V,C/4,Wf,Hf,4) like the forward's; the accumulator is fp32,
// zeroed by the caller; a layout pass (k_quad_planar_to_planar) turns it into the caller's planar gradient tensor.  Float atomics in
// the flush: run-to-run differences in the last bits (documented in the ABI).
// Autograd semantics as in the gather variant / the oracle: zero-weight taps (outside the image, z <= 0) receive nothing.
// Measured at the north-star size: r01 17.1 ms, r02 14.0, r03 12.9 per call (gather backward: 104 ms); DESIGN.md 5.2.
#include "brick_common.h"
#include "kernels.h"

namespace mvhmr {

// s_waitcnt vmcnt(0) as the builtin: the compiler's own wait-count bookkeeping sees it (an asm wait it does not, and then guards the
// grad_out registers with waits of its own in the middle of the add phase)
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); }
typedef int int4v __attribute__((ext_vector_type(4)));                 // a buffer descriptor as four SGPRs (inline asm operand)
// LDS: [ feature buffer 0 | feature buffer 1 | 4 gradient planes | BrickShared ]
//   feature buffer = kZeroBytes of zeros + cap 16-B slots (as in the forward)
//   gradient plane = (kZeroSlots + cap) floats; the first kZeroSlots only ever receive +0 (samples that are identically zero)
constexpr int kBwdLdsBytes = 160 * 1024 - 1024;
// brick shapes: 4 x (threads / 128) x 32 voxels (128-B grad_out runs per wave half), or 8 x 8 x 16 for 1024 threads (64-B runs, but a
// third fewer window pixels per voxel at ~1.3 px per voxel: the flush and the window DMA shrink with it)
constexpr int bwd_brick_x(int bz) { return bz == 32 ? kBX : 8; }
// two feature windows in LDS (the next quad's is prefetched) -- except for the 4 x 4 x 32 bricks of 8 views, which need the room for
// their windows (mean 3 300, max 4 600 slots at the configs[3] geometry; 8 x 4 x 16: 2 450 / 3 072) and keep one
constexpr int bwd_feature_buffers(int nt, int bz) { return (nt >= 1024 || bz == 16) ? 2 : 1; }
// (round 4 measured, at no gain: a second gradient plane set with the flush of quad q - 1 among the adds of quad q, planes 16 banks apart,
// half of the waves running Jacobian -> adds, two 512-thread blocks per CU -- profiles/r04_fwd_ablations.txt section K)
// slots per window set: NBUF * (kZeroBytes + 16 cap) + 4 planes * 4 B * (kZeroSlots + cap) <= the pool
constexpr int bwd_cap_slots(int nt, int bz)
{
    return ((kBwdLdsBytes - bwd_feature_buffers(nt, bz) * kZeroBytes - 16 * kZeroSlots) / (16 * bwd_feature_buffers(nt, bz) + 16)) & ~63;
}
constexpr int kAuxCmax = 12;       // BrickShared::aux word of the tap multiplicity

// Slow path of k_bwd_brick for one voxel: global float atomics per tap (bricks whose windows do not fit the LDS pool).
template <int METHOD, int VT, typename TO>
__device__ __attribute__((noinline)) void bwd_brick_slow(const float4 *fk, const TO *gobase, float *gk, const float (*proj)[12],
                                                         float c0, float c1, float c2, unsigned vox, long long N, int q_begin, int q_end, int nqv, int C, int H,
                                                         int W, int nv)
{
    // channel quads q_begin .. q_end - 1 of the nqv = (C + 3) / 4 a view holds; channels below C only (C % 4 != 0: the last quad's missing
    // channels have no grad_out)
    const int HW = H * W;
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const Taps t = make_taps(proj[v], c0, c1, c2, H, W);
        w00[v] = t.w00; w01[v] = t.w01; w10[v] = t.w10; w11[v] = t.w11;
        const int x0 = t.rx0 < 0 ? 0 : t.rx0, y0 = t.ry0 < 0 ? 0 : t.ry0;
        const int x1 = t.rx0 + 1 > W - 1 ? W - 1 : t.rx0 + 1, y1 = t.ry0 + 1 > H - 1 ? H - 1 : t.ry0 + 1;
        const int base = ((v < nv ? v : 0) * nqv) * HW;                         // an absent view reads view 0's pixels (and discards them)
        o00[v] = base + x0 * H + y0; o01[v] = base + x1 * H + y0; o10[v] = base + x0 * H + y1; o11[v] = base + x1 * H + y1;   // column-major copy
    }
    for (int q = q_begin; q < q_end; ++q) {
        const float4 *src = fk + (long long)q * HW;
        float *gq = gk + (long long)q * HW * 4;
        float s[4][VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const float4 a = src[o00[v]], bb = src[o01[v]], c = src[o10[v]], d = src[o11[v]];
            s[0][v] = bilerp(a.x, bb.x, c.x, d.x, w00[v], w01[v], w10[v], w11[v]);
            s[1][v] = bilerp(a.y, bb.y, c.y, d.y, w00[v], w01[v], w10[v], w11[v]);
            s[2][v] = bilerp(a.z, bb.z, c.z, d.z, w00[v], w01[v], w10[v], w11[v]);
            s[3][v] = bilerp(a.w, bb.w, c.w, d.w, w00[v], w01[v], w10[v], w11[v]);
            if (v >= nv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = kAbsentReads ? kAbsentSample : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (q * 4 + i >= C) break;
            const float g = to_f32<TO>(gobase[(long long)(q * 4 + i) * N + vox]);
            float ds[VT];
            if constexpr (METHOD == AGG_MEAN) aggregate_grad<AGG_SUM, VT>(s[i], __fdiv_rn(g, (float)nv), ds);   // g / (real views), as autograd of mean(0)
            else aggregate_grad<METHOD, VT>(s[i], g, ds);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                if (v >= nv) continue;                                           // absent view: nothing to receive
                if (w00[v] != 0.f) atomicAdd(gq + (long long)o00[v] * 4 + i, ds[v] * w00[v]);
                if (w01[v] != 0.f) atomicAdd(gq + (long long)o01[v] * 4 + i, ds[v] * w01[v]);
                if (w10[v] != 0.f) atomicAdd(gq + (long long)o10[v] * 4 + i, ds[v] * w10[v]);
                if (w11[v] != 0.f) atomicAdd(gq + (long long)o11[v] * 4 + i, ds[v] * w11[v]);
            }
        }
    }
}

template <int METHOD, int VT, int NT, typename TO, int BZ>
__global__ void __launch_bounds__(NT)
k_bwd_brick(const float4 *__restrict__ featK, const TO *__restrict__ grad_out, const float *__restrict__ proj,
            const Coords coords, float *__restrict__ gradK, int C, int H, int W, int X, int Y, int Z, int nby,
            int nbz, int bricks_per_sample, int lds_bytes, int total_blocks, int nv, Gate gate)
{
    // nv <= VT real views (3 views run the 4-view kernel, 5 ... 7 the 8-view one): an absent view has no camera and no window; for
    // softmax / max its samples read kAbsentSample from the zero head of the feature buffers (brick_common.h), its ds is forced to zero
    if (gated_off(gate)) return;
    // brick = BX x BY x BZ voxels, one per lane; a wave holds 64 / BZ whole z columns
    constexpr int BX = bwd_brick_x(BZ), BY = NT / (BZ * BX), NW = NT / 64, CW = 64 / BZ;
    constexpr int MC = brick_chunks_per_wave(NT);
    constexpr int NBUF = bwd_feature_buffers(NT, BZ);                               // feature windows in LDS: 2 (next quad prefetched) or 1
    constexpr int ASETS = 2;                                                        // max-|ds| words: one set per quad in flight
    extern __shared__ __align__(16) unsigned char smem[];
    BrickShared<VT> *sh = reinterpret_cast<BrickShared<VT> *>(smem + lds_bytes);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order (speed only; blocks i, i + 8, ... share an XCD): every XCD takes an eighth of every sample's brick list (an x slab).  The
    // forward's four-tiles-per-sample order (brick_common.h: brick_tiles) was measured here too: -1 % at the north star (window-fill reads 8.8 ->
    // 6.7 GB) but +3 % at configs[3] and configs[4], with either order of the bricks inside a tile: not taken (profiles/r05_bwd_ablations.txt E)
    const int share = (bricks_per_sample + 7) >> 3;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = j / share, brick = xcd * share + j % share;
    if (brick >= bricks_per_sample || b * bricks_per_sample >= total_blocks) return;
    const int kz = brick % nbz, ky = (brick / nbz) % nby, kx = brick / (nbz * nby);
    const long long N = (long long)X * Y * Z;
    const int HW = H * W, nq = C >> 2, nqv = (C + 3) >> 2;                       // nq: whole channel quads (the quad loop's); nqv: quads per view of the staged copy and the accumulator

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = tid < nv * 12 ? proj[((long long)b * nv) * 12 + tid] : 0.f;
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    __syncthreads();

    // ---- this lane's voxel and its tap records (identical to the forward)
    // ds_read_b128 serves the lanes {0-3, 12-15, 20-27} of each wave half in one pass and the other 16 in the next: z32 renumbers a
    // half so that a pass holds 16 consecutive z (of one column: distinct window rows, odd row stride -> distinct bank groups)
    const int l5 = lane & 31;
    const int z32 = l5 < 4 ? l5 : l5 < 12 ? 12 + l5 : l5 < 16 ? l5 - 8 : l5 < 20 ? 8 + l5 : l5 < 28 ? l5 - 12 : l5;
    const int kq = (lane >> 5) * 2 + (z32 >> 4);                                  // which of the wave's four columns (BZ == 16)
    const int col = BZ == 32 ? wave * 2 + (lane >> 5) : ((wave >> 2) + (BY / 2) * (kq >> 1)) * BX + (wave & 3) + 4 * (kq & 1);
    const int zin = z32 % BZ;
    // bricks past the volume's edge (extents that do not divide): such a lane idles on a clamped voxel -- no taps, zero weights, grad_out
    // read as zero (its buffer offset carries bit 31), so it adds integer zeros to its parking words and raises no maximum
    const int vx_raw = kx * BX + col % BX, vy_raw = ky * BY + col / BX, vz_raw = kz * BZ + zin;
    const bool inside = vx_raw < X && vy_raw < Y && vz_raw < Z;
    const int vx = vx_raw < X ? vx_raw : X - 1, vy = vy_raw < Y ? vy_raw : Y - 1, vz = vz_raw < Z ? vz_raw : Z - 1;
    const unsigned vox = (unsigned)(((long long)vx * Y + vy) * Z + vz);
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int tx[VT], ty[VT];
    unsigned valid = 0;
    float c0, c1, c2;
    voxel_xyz(coords, b, N, vox, c0, c1, c2);
    {
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            // The windows are COLUMN-major like the forward's and like the staged copy (B,V,C/4,W,H,4): from here on the kernel works
            // on the transposed image -- "x" is the image row index y (the fast axis of the copy), "y" the column index x, Wt = H
            // columns, Ht = W rows -- so every line below reads as the row-major algorithm it was written as.  (Windows along the
            // other axis: 13.2 instead of 13.75 ms at the north star, whose volume z axis projects onto image y; and the fused route's
            // quad-planar copy is used as it is.)
            const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
            const bool any = t.any && inside && v < nv;
            w00[v] = inside ? t.w00 : 0.f; w01[v] = inside ? t.w10 : 0.f; w10[v] = inside ? t.w01 : 0.f; w11[v] = inside ? t.w11 : 0.f;
            tx[v] = t.ry0; ty[v] = t.rx0;
            if (any) valid |= 1u << v;
            const int big = 1 << 30;
            const int xmin = wave_min(any ? t.ry0 : big), ymin = wave_min(any ? t.rx0 : big);
            const int xmax = wave_max(any ? t.ry0 : -big), ymax = wave_max(any ? t.rx0 : -big);
            if (lane == 0 && xmax >= xmin) {
                atomicMin(&sh->bbox[v][0], xmin); atomicMin(&sh->bbox[v][1], ymin);
                atomicMax(&sh->bbox[v][2], xmax); atomicMax(&sh->bbox[v][3], ymax);
            }
        }
    }
    __syncthreads();

    int wx0[VT], wy0[VT], ws[VT], bwv[VT], bhv[VT], nch[VT + 1], slot0[VT];
    nch[0] = 0;
    int used = 0, max_stride = 0;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
        const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
        int bw = 0, bh = 0;
        if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }
        const int stride = bw | 1;
        const int chunks = (stride * bh + 63) >> 6;
        wx0[v] = xmin; wy0[v] = ymin; ws[v] = stride; bwv[v] = bw; bhv[v] = bh;
        max_stride = stride > max_stride ? stride : max_stride;
        slot0[v] = used;
        used += chunks << 6;
        nch[v + 1] = nch[v] + chunks;
    }
    // capacity (bwd_cap_slots): compile-time, so that the plane and
    // buffer strides fold into the immediate offsets of the ds_ instructions (run-time strides cost 32 address registers)
    constexpr int cap = bwd_cap_slots(NT, BZ);
    constexpr int buf_bytes = kZeroBytes + cap * 16;
    constexpr int plane_floats = kZeroSlots + cap;
    int *const iplanes = reinterpret_cast<int *>(smem + NBUF * buf_bytes);
    const bool fits = used <= cap && nch[VT] <= MC * NW && max_stride + 2 <= kZeroSlots;
    const float4 *const fk = featK + (long long)b * nv * nqv * HW;
    float *const gk = gradK + (long long)b * nv * nqv * HW * 4;
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;
    const TO *const gobase = grad_out + (long long)b * C * N;
    const unsigned chan_bytes = (unsigned)(N * 4);
    const unsigned voxb = inside ? vox * 4u : 0x80000000u;                       // beyond num_records: the load returns 0

    if (fits) {
        for (int i = tid; i < kZeroSlots * NBUF; i += NT) {
            const float z = (kAbsentReads && nv < VT && i % kZeroSlots == kAbsentSlot) ? kAbsentSample : 0.f;
            *reinterpret_cast<float4 *>(smem + (i / kZeroSlots) * buf_bytes + (i % kZeroSlots) * 16) = make_float4(z, z, z, z);
        }
        for (int i = tid; i < 4 * plane_floats; i += NT) iplanes[i] = 0;
        if (tid < 13) sh->aux[tid] = 0;

        const unsigned plane0 = (unsigned)(size_t)(lds_void_t *)smem + (unsigned)(NBUF * buf_bytes);   // LDS byte address of gradient plane 0
        int a0[VT], ws16[VT];
        unsigned ga4[VT], gb4[VT];                                        // LDS byte address of the tap's two rows in gradient plane 0
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const bool ok = (valid >> v) & 1u;
            const int s0 = slot0[v] + (ty[v] - wy0[v]) * ws[v] + (tx[v] - wx0[v]);
            a0[v] = ok ? kZeroBytes + s0 * 16 : 0;
            // byte offset inside a gradient plane.  A lane whose sample is identically zero (weights 0: it adds integer zeros) is
            // parked on its own pair of words of the always-zero head (rows 0 / 1 of its "taps" then lie <= 126 + 127 words further:
            // still zero head or real slots, where + 0 changes nothing) -- parked on ONE word the lanes would serialise (same-address
            // LDS atomics: 2 cycles per lane), and masking them off instead costs an exec mask + branch per (channel, view) for nothing:
            // an exec-masked ds_add_u32 is as expensive as a full one (scripts/microbench_ldsmask.hip)
            ga4[v] = plane0 + (unsigned)(ok ? kZeroSlots + s0 : 2 * lane) * 4u;
            gb4[v] = ga4[v] + (unsigned)ws[v] * 4u;
            ws16[v] = ws[v] * 16;
            if (kAbsentReads && v >= nv) {                                       // the absent view's one "tap" (its adds are zeros: ds is forced to 0)
                a0[v] = kAbsentSlot * 16; ws16[v] = 16;
                w00[v] = 1.f; w01[v] = 0.f; w10[v] = 0.f; w11[v] = 0.f;
            }
        }
        // ---- chunks of this wave (64 consecutive window slots of one view): DMA source + flush destination
        unsigned g_off[MC];                                              // bit 31: the lane's slot is NOT a window pixel inside the image
        int l_dst[MC], c_slot[MC];
#pragma unroll
        for (int r = 0; r < MC; ++r) {
            const int c = wave + r * NW;
            l_dst[r] = -1; c_slot[r] = 0; g_off[r] = 0;
            if (c < nch[VT]) {
                int v = 0;
#pragma unroll
                for (int u = 1; u < VT; ++u) v += c >= nch[u] ? 1 : 0;
                int sv = ws[0], ox = wx0[0], oy = wy0[0], c0 = nch[0], s0 = slot0[0], bw = bwv[0], bh = bhv[0];
#pragma unroll
                for (int u = 1; u < VT; ++u) if (v == u) { sv = ws[u]; ox = wx0[u]; oy = wy0[u]; c0 = nch[u]; s0 = slot0[u]; bw = bwv[u]; bh = bhv[u]; }
                const int jj = c - c0, slot = (jj << 6) + lane;
                const int py = slot / sv, px = slot - py * sv;
                const int gx = ox + px, gy = oy + py;
                const int Wt = H, Ht = W;                                        // transposed image (see the tap records)
                const unsigned live = (px < bw && py < bh && gx >= 0 && gx < Wt && gy >= 0 && gy < Ht) ? 1u : 0u;
                const int cx = gx < 0 ? 0 : (gx > Wt - 1 ? Wt - 1 : gx), cy = gy < 0 ? 0 : (gy > Ht - 1 ? Ht - 1 : gy);
                g_off[r] = ((unsigned)((v * nqv) * HW + cy * Wt + cx) * 16u) | (live ? 0u : 0x80000000u);
                l_dst[r] = kZeroBytes + (s0 + (jj << 6)) * 16;
                c_slot[r] = kZeroSlots + s0 + (jj << 6);
            }
        }
        const unsigned lds_base = (unsigned)(size_t)(lds_void_t *)smem;
        auto dma = [&](int q) {
            const float4 *src = fk + (long long)q * HW;
            const int boff = (q & (NBUF - 1)) * buf_bytes;
#pragma unroll
            for (int r = 0; r < MC; ++r)
                if (l_dst[r] >= 0) glds16(src, g_off[r] & 0x7fffffffu, lds_base + (unsigned)uniform(l_dst[r] + boff));
        };

        // grad_out of this voxel's 4 channels (128-B runs per channel across the wave), one quad ahead
        float gn[4];
        auto load_g_to = [&](int q, float (&gn)[4]) __attribute__((always_inline)) {
            if constexpr (sizeof(TO) == 4) {
                const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<TO *>(gobase) + (long long)(q * 4) * N, 0, (int)(4u * chan_bytes), 0x00020000);
#pragma unroll
                for (int i = 0; i < 4; ++i) gn[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (int)voxb, (int)(i * chan_bytes), 0));
            } else {                                                             // 16-bit storage: 64-B runs per channel
                // plain global loads from the (clamped, always valid) voxel: buffer_load_ushort runs this kernel at 14.9 ms against 12.9.
                // gn holds the 16 raw bits; the conversion happens where the Jacobian uses them (grad_of) -- a conversion or select here
                // pins the wait for the load to this spot (13.6 ms)
                const TO *gp = gobase + (long long)(q * 4) * N + vox;
#pragma unroll
                for (int i = 0; i < 4; ++i) gn[i] = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, gp[(long long)i * N]));
            }
        };
        auto load_g = [&](int q) __attribute__((always_inline)) { load_g_to(q, gn); };
        // (the same loads as inline asm outside the compiler's wait-count bookkeeping -- no s_waitcnt vmcnt(3 - i) in front of each channel's
        // Jacobian -- were measured at no difference in round 4: fp32 13.06 vs 13.03 ms)
        auto grad_of = [&](float held) __attribute__((always_inline)) {          // the value load_g_to left in gn[i]
            if constexpr (sizeof(TO) == 4) return held;
            // (a lane past the volume's edge keeps the gradient of the edge voxel it is clamped to: its weights are zero, so it adds nothing,
            // and its ds = g / V-like values can only raise the block's scale estimate by a bounded factor.  Zeroing it here would keep the
            // `inside` mask live through the quad loop: 14.8 ms instead of 12.9 -- SGPR spills in the loop)
            else return to_f32<TO>(__builtin_bit_cast(TO, (unsigned short)__builtin_bit_cast(unsigned, held)));
        };
        dma(0);
        load_g(0);
        // ---- most taps any window pixel receives from this brick (once per brick): sizes the fixed-point headroom
        lds_barrier();                                                           // planes are zero
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            int *p = iplanes + ((ga4[v] - plane0) >> 2);
            if ((valid >> v) & 1u) { lds_add(p, 1); lds_add(p + 1, 1); lds_add(p + ws[v], 1); lds_add(p + ws[v] + 1, 1); }
        }
        lds_barrier();
        {
            int m = 1;
            for (int i = tid; i < used; i += NT) { const int c = iplanes[kZeroSlots + i]; m = c > m ? c : m; }
            m = wave_max(m);
            if (lane == 0) atomicMax(&sh->aux[kAuxCmax], m);
        }
        lds_barrier();
        // headroom: the window sums stay below 2^30 when every contribution is below 2^hbits
        const int cmax = uniform(sh->aux[kAuxCmax]);
        const int hbits = 30 - (cmax <= 1 ? 0 : 32 - __builtin_clz(cmax - 1));   // 30 - ceil(log2(cmax))
        for (int i = tid; i < plane_floats; i += NT) iplanes[i] = 0;
        wait_vm0();
        lds_barrier();
        // ---- quad loop, two barriers per quad:
        //   accumulate quad q   64 ds_add_u32 per lane, fire and forget (ds of quad q was computed one iteration earlier)
        //   (first)             re-sample quad q+1 (LDS reads ahead of the adds: the LDS pipe is in order)
        //   then                request the window / grad_out of quad q+2; Jacobian of quad q+1 -> ds, publish the block-wide
        //                       max |ds| per channel: the LDS pipe works off the adds underneath this arithmetic
        //   barrier             adds of quad q complete, max of quad q+1 known
        //   flush quad q        window -> global float atomics (256 contiguous bytes per instruction), planes zeroed
        //   barrier             planes zero, window q+2 landed (counted wait: the flush atomics stay in flight)
        // Requests go out BEFORE the flush: a CU's vector-memory pipe is in order and a flush's ~160 atomic instructions take
        // microseconds to drain -- loads queued behind them would stall the next quad.
        const unsigned aux_base = (unsigned)(size_t)(lds_void_t *)sh->aux;
        const unsigned ch4 = 4u * (lane & 3);                                    // this lane's channel in the flush
        const unsigned long long gk_bits = (unsigned long long)(size_t)gk;
        const int4v dgk = {uniform((int)(unsigned)gk_bits), uniform((int)((unsigned)(gk_bits >> 32) & 0xffffu)), (int)((unsigned)nv * nqv * HW * 16u), 0x00020000};
        const __amdgpu_buffer_rsrc_t rgk = __builtin_amdgcn_make_buffer_rsrc(gk, 0, (int)((unsigned)nv * nqv * HW * 16u), 0x00020000);   // this sample's accumulator
        float ds[4][VT], s[4][VT];
        auto resample = [&](int q) {                                             // samples of quad q from its window buffer
            const int boff = (q & (NBUF - 1)) * buf_bytes;
#pragma unroll
            for (int v = 0; v < VT; ++v) {                                       // one view at a time: 16 tap registers, not 64
                __builtin_amdgcn_sched_barrier(0);
                const int base = a0[v] + boff, row1 = base + ws16[v];
                f32x4 ta, tb, tc, td;
                ta = lds_tap(smem, base); tb = lds_tap(smem, base + 16); tc = lds_tap(smem, row1); td = lds_tap(smem, row1 + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[i][v] = bilerp(ta.v[i], tb.v[i], tc.v[i], td.v[i], w00[v], w01[v], w10[v], w11[v]);
                    asm volatile("" : "+v"(s[i][v]));                              // fold here (not sunk below the adds)
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // absent views receive nothing; the block-wide max |ds| of the channel is published for the quad's fixed-point scale
        auto publish_max = [&](int aset, int i, float (&dsi)[VT]) __attribute__((always_inline)) {
            if (nv < VT) {                                                       // wave-uniform, as is every test below
#pragma unroll
                for (int v = 1; v < VT; ++v)                                     // a move, not a product: 0 * fma(g, -FLT_MAX, c) can be NaN.  (As
                    if (v >= nv) asm volatile("v_mov_b32 %0, 0" : "=v"(dsi[v])); // selects the compiler keeps VT lane masks in SGPR pairs across the loop)
            }
            // max |ds| on the BITS (sign cleared): non-negative floats order as ints, and Inf / NaN (>= 0x7f800000) sort above
            // every finite value instead of being dropped as fmaxf drops a NaN -- a non-finite gradient must stay visible
            int big = 0;
            if constexpr (VT == 4) {
                const int b0 = __builtin_bit_cast(int, dsi[0]) & 0x7fffffff, b1 = __builtin_bit_cast(int, dsi[1]) & 0x7fffffff;
                const int b2 = __builtin_bit_cast(int, dsi[2]) & 0x7fffffff, b3 = __builtin_bit_cast(int, dsi[3]) & 0x7fffffff;
                asm("v_max3_i32 %0, %1, %2, %3" : "=v"(big) : "v"(b0), "v"(b1), "v"(b2));
                big = b3 > big ? b3 : big;
            } else {
#pragma unroll
                for (int v = 0; v < VT; ++v) { const int a = __builtin_bit_cast(int, dsi[v]) & 0x7fffffff; big = a > big ? a : big; }
            }
            // one scale per channel of the quad: wave max -> lane 63 -> ds_max into the block's word
            lds_max_from_lane63(aux_base + (unsigned)((aset * 4 + i) * 4), wave_max_to_last_row(big));
        };
        // ds of channel i of quad q from its samples and gn[i]
        auto jacobian_to = [&](int aset, int i, float (&dsi)[VT]) __attribute__((always_inline)) {
            const float gi = grad_of(gn[i]);
            if constexpr (METHOD == AGG_MEAN) aggregate_grad<AGG_SUM, VT>(s[i], __fdiv_rn(gi, (float)nv), dsi);   // g / (real views), as autograd of mean(0)
            else aggregate_grad<METHOD, VT>(s[i], gi, dsi);
            publish_max(aset, i, dsi);
        };
        auto jacobian1 = [&](int aset, int i) __attribute__((always_inline)) { jacobian_to(aset, i, ds[i]); };
        if (NBUF == 2 && nq > 1) dma(1);
        resample(0);
        if (NBUF == 1 && nq > 1) { lds_barrier(); dma(1); }                      // single buffer: every wave has sampled window 0
#pragma unroll
        for (int i = 0; i < 4; ++i) jacobian1(0, i);
        if (nq > 1) load_g(1);
        int aset = 0;                                                            // aux set of quad q (q % ASETS)
        wait_vm0();                                                              // window 1 has landed
        lds_barrier();
#pragma nounroll
        for (int q = 0; q < nq; ++q) {
            // LDS float atomics run at ~190 cycles per wave instruction on gfx950, integer ones at ~4-6, so the window is
            // accumulated in fixed point: contributions are scaled by a power of two chosen per (brick, quad) from the
            // block-wide max |ds| (weights are <= 1) and the brick's tap multiplicity, rounded to int32, ds_add_u32.
            // 2^e > max |ds| of the channel  ->  scale = 2^(hbits - e): |ds * w * scale| < 2^hbits.  Exponent clamped to normal
            // floats; an all-zero channel adds nothing.
            // The LDS pipe is in order: the window reads of quad q+1 go ahead of this quad's 64 adds per lane, whose
            // service time then hides under the Jacobian of quad q+1 (VALU only).
            if (q + 1 < nq) resample(q + 1);
            // scales: integer arithmetic on wave-uniform values, kept on the scalar unit (the asm operands below are SGPRs) -- as float
            // selects the compiler built them with ~27 half-rate VALU instructions per quad
            float scale[4];
            int inv_bits[4];
            bool poisoned = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bbits = uniform(sh->aux[aset * 4 + i]);
                const int e = ((bbits >> 23) & 0xff) - 126;
                int se = hbits - e + 127;
                se = se < 1 ? 1 : (se > 254 ? 254 : se);
                int sc_bits = bbits == 0 ? 0 : se << 23;
                inv_bits[i] = bbits == 0 ? 0 : (254 - se) << 23;
                // a non-finite |ds| somewhere in the brick (overflowed or NaN grad_out): fixed point cannot carry it.  The channel
                // adds nothing (scale 0) and the flush writes NaN to every live pixel of the brick's windows instead -- a superset
                // of the pixels the reference's float scatter would poison, so that isfinite() checks downstream still trip.
                if (bbits >= 0x7f800000) { sc_bits = 0; inv_bits[i] = 0x7fc00000; poisoned = true; }
                asm volatile("v_mov_b32 %0, %1" : "=v"(scale[i]) : "s"(sc_bits));
            }
            if (q + 2 < nq) {
                if (NBUF == 1) lds_barrier();                                    // single buffer: every wave has sampled window q+1
                dma(q + 2);                                                      // into the buffer quad q (or q+1) was sampled from
            }
            const int aset_next = aset + 1 == ASETS ? 0 : aset + 1;
            // channel by channel: the 16 adds of quad q (fire and forget: the LDS unit retires ~14.5 lane-adds per clock, so a wave
            // issues them at the pace of the whole CU's queue), then the Jacobian of quad q+1 for the same channel (VALU only; it
            // overwrites ds[i], which the adds just issued have read) -- the arithmetic runs while the queue drains
            // Address operands: the two row addresses of a view are per-lane registers kept for the whole kernel (plane 0, absolute LDS
            // bytes); the plane and the +1 column are the instruction's immediate offset.  No VALU address arithmetic per add -- a
            // v_add_u32 with an SGPR operand (the plane base, the row stride) issues at half rate, as does the shift that rebuilt
            // the byte offset (scripts/microbench_ops.hip).
            auto adds_of_channel = [&](auto i_tag) __attribute__((always_inline)) {
                constexpr int i = decltype(i_tag)::value;
                constexpr int off = i * plane_floats * 4;
                static_assert(3 * plane_floats * 4 + 4 < 65536, "plane offsets must fit the ds_ immediate");
                // the channel's 4 VT adds of quad q (two VALU instructions + the ds_add_u32 each), then its Jacobian of quad q + 1, which
                // overwrites ds[i].  (Round 5 interleaved the two -- one add per three or four Jacobian instructions, on the theory that a
                // wave sitting in the add burst cannot issue arithmetic -- and measured nothing, 12.87 vs 12.88 ms on one box: with sixteen
                // waves at different points of the channel the other waves' arithmetic already fills the burst.  The interleaved form is
                // scripts/exp/patches/bwd_interleaved_adds.patch; profiles/r05_bwd_ablations.txt has the numbers.)
#pragma unroll
                for (int v = 0; v < VT; ++v) {
                    const float d = ds[i][v] * scale[i];
                    lds_add_at<off>(ga4[v], round_int(d * w00[v]));
                    lds_add_at<off + 4>(ga4[v], round_int(d * w01[v]));
                    lds_add_at<off>(gb4[v], round_int(d * w10[v]));
                    lds_add_at<off + 4>(gb4[v], round_int(d * w11[v]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < nq) jacobian1(aset_next, i);
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- flush: 16 window slots x 4 channels = 256 contiguous bytes of the accumulator per wave instruction.
            // A slot that is not a live pixel (padding, outside the image) carries bit 31 in its offset: beyond the buffer's
            // num_records (< 2^31, brick_bwd_supported), so the hardware drops that lane's atomic.  Lanes whose sum is exactly zero
            // are masked off with v_cmpx (zero-lane atomics cost 0.35 ms of L2 atomic time at the north star) and an instruction
            // without lanes is skipped; the wave counts what it issued (n_dyn) for the vmcnt wait below.  A quad with a poisoned
            // channel (inv_ch is NaN: 0 * NaN = NaN) adds every lane instead: NaN goes to every LIVE pixel.
            // The first form of this flush tested `add` in C++: a compare, a select, a ballot and two branches per instruction,
            // serialised behind the LDS read they test, plus the 64-bit address arithmetic of a global_atomic -- 7 slow + 3 fast
            // VALU instructions per element against 2 + 2 here (scripts/loop_histogram.py).
            // 1 / scale of this lane's channel (lane & 3): four scalar values into one register under exec masks (written as a
            // select chain hipcc turns it into a private array in scratch, indexed by lane & 3)
            float inv_ch;
            {
                unsigned long long save;
                asm volatile("v_mov_b32 %0, %2\n\t"
                             "s_mov_b64 %1, exec\n\t"
                             "s_mov_b64 exec, %6\n\tv_mov_b32 %0, %3\n\t"
                             "s_mov_b64 exec, %7\n\tv_mov_b32 %0, %4\n\t"
                             "s_mov_b64 exec, %8\n\tv_mov_b32 %0, %5\n\t"
                             "s_mov_b64 exec, %1"
                             : "=&v"(inv_ch), "=&s"(save)
                             : "s"(inv_bits[0]), "s"(inv_bits[1]), "s"(inv_bits[2]), "s"(inv_bits[3]),
                               "s"(0x2222222222222222ull), "s"(0x4444444444444444ull), "s"(0x8888888888888888ull));
            }
            int n_dyn = 0;                                                       // atomic instructions this wave issues in this iteration
            // flush of quad qf from its plane set (scaled back by inv: this lane's channel), planes left zero
            auto flush_quad = [&](int qf, float inv, bool pois) __attribute__((always_inline)) {
                const int q_off = qf * HW * 16;                                  // wave-uniform byte offset of the quad (soffset)
                int *const pset = iplanes;
                auto flush = [&](auto masked_tag) __attribute__((always_inline)) {
                    constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
                    for (int r = 0; r < MC; ++r) {
                        if (l_dst[r] < 0) continue;
                        int *pl = pset + (lane & 3) * plane_floats + c_slot[r] + (lane >> 2);
                        unsigned off[4];
                        int iv[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {                         // the chunk's LDS traffic first, then its four atomics
                            off[jj] = (unsigned)__shfl((int)g_off[r], 16 * jj + (lane >> 2));   // byte offset of that slot's pixel (view base included)
                            iv[jj] = pl[16 * jj];
                            pl[16 * jj] = 0;                                     // ready for the quad after next
                        }
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const float val = (float)iv[jj] * inv;
                            const int voff = (int)(off[jj] + ch4);
                            if constexpr (MASKED) {
                                unsigned long long save;
                                asm volatile("s_mov_b64 %[save], exec\n\t"
                                             "v_cmpx_ne_u32_e32 vcc, 0, %[iv]\n\t"
                                             "s_cbranch_execz .Lskip%=\n\t"
                                             "buffer_atomic_add_f32 %[val], %[voff], %[rsrc], %[soff] offen\n\t"
                                             "s_add_u32 %[cnt], %[cnt], 1\n"
                                             ".Lskip%=:\n\t"
                                             "s_mov_b64 exec, %[save]"
                                             : [save] "=&s"(save), [cnt] "+s"(n_dyn)
                                             : [iv] "v"(iv[jj]), [val] "v"(val), [voff] "v"(voff), [rsrc] "s"(dgk), [soff] "s"(q_off)
                                             : "vcc", "scc", "memory");
                            } else {
                                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(val, rgk, voff, q_off, 0);
                                ++n_dyn;
                            }
                        }
                    }
                };
                if (pois) flush(std::false_type{}); else flush(std::true_type{});
            };
            adds_of_channel(std::integral_constant<int, 0>{});
            adds_of_channel(std::integral_constant<int, 1>{});
            adds_of_channel(std::integral_constant<int, 2>{});
            adds_of_channel(std::integral_constant<int, 3>{});
            if (q + 2 < nq) load_g(q + 2);
            lds_barrier();                                                       // adds of quad q landed; max of quad q+1 published
            flush_quad(q, inv_ch, poisoned);
            if (tid < 4) sh->aux[aset * 4 + tid] = 0;                            // read by every wave before the barrier above
            // window q+2 and grad_out q+2 (requested before this quad's atomics) have landed; the atomics stay in flight
            wait_vmcnt(n_dyn);
            lds_barrier();
            aset = aset_next;
        }
    } else {
        // ---- windows do not fit: scatter straight to the accumulator (its own function: keeps its registers -- 16 tap
        // offsets on top of the weights -- out of the fast path's allocation, which otherwise spills in the quad loop)
        if (inside) bwd_brick_slow<METHOD, VT, TO>(fk, gobase, gk, sh->proj, c0, c1, c2, vox, N, 0, nq, nqv, C, H, W, nv);
    }
}

// fp32 COLUMN-major quad-planar accumulator (BV, C/4, W, H, 4) -> planar gradient (BV, C, H, W).  One block = a band of 32 image
// columns of one channel quad, read linearly (the band is contiguous: 32 * H float4) and turned through LDS; written as 128-B runs
// along x per channel and row.
template <typename TF, int BW>
__global__ void __launch_bounds__(512)
k_quad_planar_to_planar(const float4 *__restrict__ src, TF *__restrict__ dst, int C, int H, int W, Gate gate)
{
    if (gated_off(gate)) return;
    extern __shared__ float tile[];                                              // [4][H][BW + 1]
    constexpr int TS = BW + 1;
    const long long bv = blockIdx.z;
    const int q = blockIdx.y, x0 = blockIdx.x * BW;
    const int cols = W - x0 < BW ? W - x0 : BW;
    const float4 *s = src + ((bv * ((C + 3) >> 2) + q) * W + x0) * (long long)H;
    for (int i = threadIdx.x; i < cols * H; i += 512) {
        const float4 g = s[i];
        const int xl = i / H, y = i - xl * H;
        float *t = tile + y * TS + xl;
        t[0] = g.x; t[H * TS] = g.y; t[2 * H * TS] = g.z; t[3 * H * TS] = g.w;
    }
    __syncthreads();
    TF *d = dst + (bv * C + q * 4) * (long long)H * W + x0;
    const int xl = threadIdx.x & (BW - 1);
    const int nc = C - q * 4 < 4 ? C - q * 4 : 4;                                // C % 4 != 0: the last quad's missing channels are not written
    if (xl < cols)
        for (int r = threadIdx.x / BW; r < nc * H; r += 512 / BW) d[(long long)r * W + xl] = from_f32<TF>(tile[r * TS + xl]);   // r = channel * H + y
}

// band width of the gradient layout pass: 32 columns where 4 x H x 33 floats fit 64 KB of LDS (H <= 124), else 8
inline int grad_band(const Problem &p) { return (size_t)4 * p.H * 33 * sizeof(float) <= 64 * 1024 ? 32 : 8; }

namespace {
constexpr int kNTb = 1024;                            // 2 / 4 views
constexpr int kNTb8 = 512;                            // 8 views: 4 x 4 x 32 bricks, 256 VGPRs per lane, ONE feature window in LDS

// z extent of the bricks: 8 x 8 x 16 (2 / 4 views) or 8 x 4 x 16 (8 views) when the volume divides (fp16 grad_out then comes in
// 32-B runs: still hidden, 18.1 -> 15.1 ms like fp32), 4 x BY x 32 otherwise
int bwd_brick_z(const Problem &p)
{
    const int by = p.V > 4 ? 4 : 8, nt = p.V > 4 ? kNTb8 : kNTb;
    if (p.X % 8 == 0 && p.Y % by == 0 && p.Z % 16 == 0) return 16;
    const int by32 = nt / 128;
    if (p.X % kBX == 0 && p.Y % by32 == 0 && p.Z % kBZ == 0) return kBZ;
    // neither shape divides the volume: the 16-deep bricks (a third fewer window pixels per voxel, and windows that fit where the
    // 32-deep ones overflow: 50^3 at the north-star maps 41 ms with 4 x 8 x 32 against 28 ms on the plane kernels) unless they
    // would leave over a quarter more lanes idle than the 32-deep ones
    auto cover = [](int n, int b) { return (long long)((n + b - 1) / b) * b; };
    const long long c16 = cover(p.X, 8) * cover(p.Y, by) * cover(p.Z, 16), c32 = cover(p.X, kBX) * cover(p.Y, by32) * cover(p.Z, kBZ);
    return 4 * c16 <= 5 * c32 ? 16 : kBZ;
}

template <int METHOD, int VT, int NT, typename TO, int BZ = kBZ>
hipError_t launch_bv(const float4 *featK, const TO *grad_out, const float *proj, const Coords &coords, float *gradK, const Problem &p,
                     hipStream_t s)
{
    constexpr int BX = bwd_brick_x(BZ), BY = NT / (BZ * BX);
    const int nbx = (p.X + BX - 1) / BX, nby = (p.Y + BY - 1) / BY, nbz = (p.Z + BZ - 1) / BZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int lds_bytes = kBwdLdsBytes;
    const size_t lds = (size_t)lds_bytes + sizeof(BrickShared<VT>);
    auto kern = k_bwd_brick<METHOD, VT, NT, TO, BZ>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int grid = ((bps + 7) / 8) * 8 * p.B;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, featK, grad_out, proj, coords, gradK, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps,
                       lds_bytes, total, p.V, make_gate(p, true));
    return hipGetLastError();
}

template <int METHOD, typename TO>
hipError_t launch_bm(const float4 *featK, const TO *grad_out, const float *proj, const Coords &coords, float *gradK, const Problem &p,
                     hipStream_t s)
{
    switch (brick_view_slots(p.V)) {                                       // 3 / 5 / 6 / 7 views: the next larger kernel, missing views absent
    case 2:
        if (bwd_brick_z(p) == 16) return launch_bv<METHOD, 2, kNTb, TO, 16>(featK, grad_out, proj, coords, gradK, p, s);
        return launch_bv<METHOD, 2, kNTb, TO>(featK, grad_out, proj, coords, gradK, p, s);
    case 4:
        if (bwd_brick_z(p) == 16) return launch_bv<METHOD, 4, kNTb, TO, 16>(featK, grad_out, proj, coords, gradK, p, s);
        return launch_bv<METHOD, 4, kNTb, TO>(featK, grad_out, proj, coords, gradK, p, s);
    case 8:
        if (bwd_brick_z(p) == 16) return launch_bv<METHOD, 8, kNTb8, TO, 16>(featK, grad_out, proj, coords, gradK, p, s);
        return launch_bv<METHOD, 8, kNTb8, TO>(featK, grad_out, proj, coords, gradK, p, s);
        break;
    }
    return hipErrorNotSupported;
}

template <typename TO>
hipError_t launch_bt(const float4 *fk, const TO *go, const float *proj, const Coords &coords, float *gradK, const Problem &p, hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return launch_bm<AGG_SOFTMAX, TO>(fk, go, proj, coords, gradK, p, s);
    case AGG_SUM: return launch_bm<AGG_SUM, TO>(fk, go, proj, coords, gradK, p, s);
    case AGG_MEAN: return launch_bm<AGG_MEAN, TO>(fk, go, proj, coords, gradK, p, s);
    case AGG_MAX: return launch_bm<AGG_MAX, TO>(fk, go, proj, coords, gradK, p, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace

// ---- C % 4 != 0: k_bwd_brick's quad loop runs the C / 4 whole channel quads; the last, partial quad goes per voxel through bwd_brick_slow
// (float atomics into the accumulator: those 1 ... 3 channels' gradient is not bit-reproducible between runs), one thread per voxel, launched
// behind the brick kernel.  (As a cold tail inside k_bwd_brick the same code would keep kernel arguments alive through the quad loop.)
template <int METHOD, int VT, typename TO>
__global__ void __launch_bounds__(256)
k_bwd_tail(const float4 *__restrict__ featK, const TO *__restrict__ grad_out, const float *__restrict__ proj, const Coords coords, float *__restrict__ gradK,
           int C, int H, int W, long long N, int nv, Gate gate)
{
    if (gated_off(gate)) return;
    __shared__ float sproj[VT][12];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < VT * 12) sproj[tid / 12][tid % 12] = tid < nv * 12 ? proj[((long long)b * nv) * 12 + tid] : 0.f;
    __syncthreads();
    const long long n = (long long)blockIdx.x * 256 + tid;
    if (n >= N) return;
    const int nqv = (C + 3) >> 2;
    const long long HW = (long long)H * W;
    float c0, c1, c2;
    voxel_xyz(coords, b, N, (unsigned)n, c0, c1, c2);
    bwd_brick_slow<METHOD, VT, TO>(featK + (long long)b * nv * nqv * HW, grad_out + (long long)b * C * N, gradK + (long long)b * nv * nqv * HW * 4, sproj, c0, c1, c2,
                                   (unsigned)n, N, C >> 2, nqv, nqv, C, H, W, nv);
}

namespace {
template <int METHOD, typename TO>
hipError_t launch_bwd_tail_views(const float4 *fk, const TO *go, const float *proj, const Coords &coords, float *gradK, const Problem &p, hipStream_t s)
{
    const dim3 grid((unsigned)((p.N + 255) / 256), (unsigned)p.B);
    const Gate gate = make_gate(p, true);
    switch (brick_view_slots(p.V)) {
    case 2: hipLaunchKernelGGL((k_bwd_tail<METHOD, 2, TO>), grid, dim3(256), 0, s, fk, go, proj, coords, gradK, p.C, p.H, p.W, p.N, p.V, gate); break;
    case 4: hipLaunchKernelGGL((k_bwd_tail<METHOD, 4, TO>), grid, dim3(256), 0, s, fk, go, proj, coords, gradK, p.C, p.H, p.W, p.N, p.V, gate); break;
    case 8: hipLaunchKernelGGL((k_bwd_tail<METHOD, 8, TO>), grid, dim3(256), 0, s, fk, go, proj, coords, gradK, p.C, p.H, p.W, p.N, p.V, gate); break;
    default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

template <typename TO>
hipError_t launch_bwd_tail(const float4 *fk, const TO *go, const float *proj, const Coords &coords, float *gradK, const Problem &p, hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return launch_bwd_tail_views<AGG_SOFTMAX, TO>(fk, go, proj, coords, gradK, p, s);
    case AGG_SUM: return launch_bwd_tail_views<AGG_SUM, TO>(fk, go, proj, coords, gradK, p, s);
    case AGG_MEAN: return launch_bwd_tail_views<AGG_MEAN, TO>(fk, go, proj, coords, gradK, p, s);
    case AGG_MAX: return launch_bwd_tail_views<AGG_MAX, TO>(fk, go, proj, coords, gradK, p, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace

// featK: quad-planar features; gradK: zeroed fp32 quad-planar accumulator of the same shape
hipError_t launch_bwd_brick(const void *featK, const void *grad_out, const float *proj, const Coords &coords, float *gradK, const Problem &p,
                            hipStream_t s)
{
    if (!brick_bwd_supported(p)) return hipErrorNotSupported;
    const float4 *fk = static_cast<const float4 *>(featK);
    hipError_t e;
    if (p.out_bf16) e = launch_bt<bf16_t>(fk, static_cast<const bf16_t *>(grad_out), proj, coords, gradK, p, s);
    else e = p.out_f16 ? launch_bt<__half>(fk, static_cast<const __half *>(grad_out), proj, coords, gradK, p, s)
                       : launch_bt<float>(fk, static_cast<const float *>(grad_out), proj, coords, gradK, p, s);
    if (e != hipSuccess || !(p.C & 3)) return e;
    if (p.out_bf16) return launch_bwd_tail<bf16_t>(fk, static_cast<const bf16_t *>(grad_out), proj, coords, gradK, p, s);   // the last, partial quad
    return p.out_f16 ? launch_bwd_tail<__half>(fk, static_cast<const __half *>(grad_out), proj, coords, gradK, p, s)
                     : launch_bwd_tail<float>(fk, static_cast<const float *>(grad_out), proj, coords, gradK, p, s);
}

hipError_t launch_quad_grad_to_planar(const float *gradK, void *dst, const Problem &p, hipStream_t s)
{
    const int bw = grad_band(p);
    const dim3 grid((p.W + bw - 1) / bw, p.C4 / 4, p.B * p.V);
    const size_t lds = (size_t)4 * p.H * (bw + 1) * sizeof(float);
    const float4 *g = (const float4 *)gradK;
    const Gate gate = make_gate(p, true);
    if (bw == 32) {
        if (p.feat_f16) hipLaunchKernelGGL((k_quad_planar_to_planar<__half, 32>), grid, dim3(512), lds, s, g, (__half *)dst, p.C, p.H, p.W, gate);
        else hipLaunchKernelGGL((k_quad_planar_to_planar<float, 32>), grid, dim3(512), lds, s, g, (float *)dst, p.C, p.H, p.W, gate);
    } else {
        if (p.feat_f16) hipLaunchKernelGGL((k_quad_planar_to_planar<__half, 8>), grid, dim3(512), lds, s, g, (__half *)dst, p.C, p.H, p.W, gate);
        else hipLaunchKernelGGL((k_quad_planar_to_planar<float, 8>), grid, dim3(512), lds, s, g, (float *)dst, p.C, p.H, p.W, gate);
    }
    return hipGetLastError();
}

bool brick_bwd_supported(const Problem &p)
{
    if ((p.out_f16 && !p.feat_f16) || (p.out_bf16 && p.feat_f16)) return false;   // grad_out and the feature gradient each in their own storage type; these two pairings do not exist (capi: check_desc)
    if (p.V < 1 || p.V > 8) return false;
    if (p.C < 4 || ((p.C & 3) && p.B > 65535)) return false;                      // r05: C % 4 != 0 -- whole quads in the quad loop, the rest per voxel (k_bwd_tail: grid.y = B)
    if ((long long)p.B * p.V * (p.C4 / 4) * p.H * p.W >= (1ll << 31)) return false;
    if ((long long)p.V * (p.C4 / 4) * p.H * p.W >= (1ll << 27)) return false;   // one sample's accumulator: 32-bit byte offsets (buffer atomics)
    if (p.N >= (1ll << 28)) return false;
    if ((size_t)4 * p.H * 9 * sizeof(float) > 64 * 1024) return false;           // the gradient layout pass turns column bands through LDS (H <= 455)
    return true;
}

GateGeom brick_bwd_gate_geom(const Problem &p)
{
    const int nt = p.V > 4 ? kNTb8 : kNTb;
    GateGeom g;
    g.bz = bwd_brick_z(p);
    g.bx = bwd_brick_x(g.bz); g.by = nt / (g.bz * g.bx); g.column_major = 1; g.view_group = 0; g.parity_rows = 0;
    g.cap_slots = bwd_cap_slots(nt, g.bz);
    g.max_chunks = brick_chunks_per_wave(nt) * (nt / 64);
    return g;
}

}  // namespace mvhmr
