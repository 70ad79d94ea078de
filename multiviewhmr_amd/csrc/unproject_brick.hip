// "brick" variant of the fused un-projection forward: LDS-staged feature patches.
//
// Why.  Per (voxel, view, channel) the sampler moves 4 taps x 4 B from the memory system into registers:
// 137 GB at the north-star size (64^3 x 256 ch x 4 views x 32 samples) against 9.9 GB of compulsory HBM
// traffic.  Through the vector L1 (64 B/clk/CU, ~17 TB/s chip-wide for gathers) that alone costs ~8 ms --
// the gather variant's time.  LDS delivers 256 B/clk/CU to ds_read_b128, so the taps must come from LDS,
// and every feature pixel must be fetched from L2 far less often than it is sampled.
//
// Mapping (CDNA4, wave64).
//   block  = one voxel brick of 4 x (NT/128) x 32 voxels of one sample, NT threads (1024 for 2 / 4 views, 512 for 8
//            views: 256 VGPRs per lane); a lane owns ONE voxel for the block's lifetime, so its tap records (LDS
//            address, 4 weights per view) are computed ONCE (device_common.h::make_taps, reference
//            aggregation.py:38-54) and stay in registers while the block loops over all channel quads;
//   z-long bricks: the (B,C,X,Y,Z) output is written in runs of 32 consecutive z = 128 B per (channel,
//            column) -- measured floor for full-rate HBM writes on MI355X (64-B runs: 3.4 TB/s, 32-B: 0.7);
//   layout = features are re-laid "quad-planar" (B,V,C/4,Hf,Wf,4) by a pre-pass, so a pixel's 4 channels
//            are one 16-B LDS slot and a window row is one contiguous global segment;
//   window = per view, the bounding box of the brick's taps (wave shuffles + one LDS atomic per wave);
//            the views' windows are packed back to back in one LDS pool (a brick near one camera of the
//            ring is far from the opposite one, so the SUM of the windows is what has to fit);
//   staging = LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), per channel quad, into a ring of 3 buffers
//            (2 when the windows need the room): the DMA runs three quads ahead; one bare s_barrier per quad behind
//            a counted s_waitcnt vmcnt(1) that leaves the latest output store in flight;
//   loop   = per view: fold the taps of quad q (bilinear FMAs) and request the same view of quad q+1 at once, so that
//            LDS reads are queued during every phase of the iteration;
//   storage = fp32, or fp16 features / volume with the staged copy kept fp32 (widened once by the layout pass);
//   bricks whose windows do not fit the pool (exotic cameras, huge maps) take a slower block-uniform path
//            that samples from global memory, so geometry can only cost speed, never correctness;
//   zero padding = a tap outside the image has weight 0 (make_taps) and its staged pixel is clamped into
//            the image, i.e. contributes 0 * finite (aggregation.py:55-58, padding_mode='zeros').
// The cross-view aggregate runs in registers exactly as in the gather variant (aggregation.py:71-85).
#include "brick_common.h"
#include "kernels.h"

namespace mvhmr {

// features (BV, C, HW) fp32 or fp16 -> fp32 (BV, C/4, HW, 4): the staged copy is fp32 for either storage type (the LDS-DMA
// moves raw bytes; fp16 features cost one widening here instead of 16 conversions per voxel, view and quad in the loop)
template <typename TF>
__global__ void __launch_bounds__(256)
k_to_quad_planar(const TF *__restrict__ src, float4 *__restrict__ dst, int C, int HW, Gate gate)
{
    if (gated_off(gate)) return;
    const long long bv = blockIdx.z;
    const int q = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const TF *s = src + (bv * C + q * 4) * HW + p;
    dst[(bv * (C >> 2) + q) * HW + p] = make_float4(to_f32<TF>(s[0]), to_f32<TF>(s[HW]), to_f32<TF>(s[2 * (long long)HW]), to_f32<TF>(s[3 * (long long)HW]));
}

// NT threads, brick = 4 x (NT/128) x 32 voxels, one voxel per lane.
// LDS: [ ring of 2 or 3 buffers | BrickShared ]; a buffer = a zero region + the views' windows, 16-B slots.
template <int METHOD, int VT, int NT, typename TO>
__global__ void __launch_bounds__(NT)
k_fwd_brick(const float4 *__restrict__ featK, const float *__restrict__ proj, const float *__restrict__ coords,
            TO *__restrict__ out, int C, int H, int W, int X, int Y, int Z, int nby, int nbz, int bricks_per_sample,
            int lds_slots, int total_blocks, Gate gate)
{
    if (gated_off(gate)) return;
    constexpr int BY = NT / 128, NW = NT / 64;
    constexpr int MC = brick_chunks_per_wave(NT);                                 // DMA chunks a wave may own per quad
    extern __shared__ __align__(16) unsigned char smem[];
    BrickShared<VT> *sh = reinterpret_cast<BrickShared<VT> *>(smem + lds_slots * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order.  Blocks i, i+8, ... share an XCD (round-robin dispatch; placement only affects speed).  All eight
    // XCDs work on the SAME sample at a time, each on a contiguous eighth of its bricks: neighbouring windows meet in one
    // L2, and the feature planes live across the chip are one or two samples (37.7 MB each at the north-star size) -- they
    // stay in the 256 MB Infinity Cache instead of being re-fetched from HBM (8 samples at once did not fit: 7.9 GB read).
    // Each XCD takes a compact tile of brick COLUMNS (all z): tiles_x x tiles_y = 8 tiles over the (x, y) brick grid.  A
    // compact tile projects to a compact image region in every view, so an XCD pulls ~1/3 of each feature plane through
    // its L2 instead of most of it (an x-slab covers the whole image for cameras that look along y).
    const int nbx = bricks_per_sample / (nby * nbz);
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int tw = (nbx + tiles_x - 1) / tiles_x, th = (nby + tiles_y - 1) / tiles_y;   // tile size in brick columns
    const int share = tw * th * nbz;                                             // work items of one sample per XCD
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = j / share, r = j % share;
    const int kz = r % nbz, cy = (r / nbz) % th, cx = r / (nbz * th);
    const int kx = (xcd % tiles_x) * tw + cx, ky = (xcd / tiles_x) * th + cy;
    if (kx >= nbx || ky >= nby || b * bricks_per_sample >= total_blocks) return;
    const long long N = (long long)X * Y * Z;
    const int HW = H * W, nq = C >> 2;

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = proj[((long long)b * VT) * 12 + tid];
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    __syncthreads();

    // ---- this lane's voxel and its tap records (once per brick)
    const int col = wave * 2 + (lane >> 5);
    // z of this lane inside the brick: the four 16-lane groups ds_read_b128 services one after the other are
    // {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32; give each of them 16 CONSECUTIVE z so that a group spans
    // ~23 window rows instead of ~40 (fewer slot collisions); the 32 lanes still fill one 128-B run of every store
    const int l5 = lane & 31;
    const int zin = l5 < 4 ? l5 : l5 < 12 ? 12 + l5 : l5 < 16 ? l5 - 8 : l5 < 20 ? 8 + l5 : l5 < 28 ? l5 - 12 : l5;
    const int vx = kx * kBX + (col & 3), vy = ky * BY + (col >> 2), vz = kz * kBZ + zin;
    const unsigned vox = (unsigned)(((long long)vx * Y + vy) * Z + vz);              // N < 2^30 (brick_supported)
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int tx[VT], ty[VT];
    unsigned valid = 0;
    {
        const float *Xp = coords + ((long long)b * N + vox) * 3;
        const float c0 = Xp[0], c1 = Xp[1], c2 = Xp[2];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
            w00[v] = t.w00; w01[v] = t.w01; w10[v] = t.w10; w11[v] = t.w11;
            tx[v] = t.rx0; ty[v] = t.ry0;
            if (t.any) valid |= 1u << v;
            const int big = 1 << 30;
            const int xmin = wave_min(t.any ? t.rx0 : big), ymin = wave_min(t.any ? t.ry0 : big);
            const int xmax = wave_max(t.any ? t.rx0 : -big), ymax = wave_max(t.any ? t.ry0 : -big);
            if (lane == 0 && xmax >= xmin) {
                atomicMin(&sh->bbox[v][0], xmin); atomicMin(&sh->bbox[v][1], ymin);
                atomicMax(&sh->bbox[v][2], xmax); atomicMax(&sh->bbox[v][3], ymax);
            }
        }
    }
    __syncthreads();

    // ---- window per view (block-uniform): origin, width, odd row stride, rows, first slot; views packed back to back
    int wx0[VT], wy0[VT], ws[VT], nch[VT + 1], slot0[VT];
    nch[0] = 0;
    int used = 0, max_stride = 0;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
        const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
        int bw = 0, bh = 0;
        if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }     // taps reach x0+1, y0+1
        // row stride in 16-B slots, forced odd: the 16 lanes of a ds_read_b128 group are 16 consecutive voxels of
        // a column, i.e. ~16 different window rows; an odd stride spreads them over all 16 slots of the bank row
        const int stride = bw | 1;
        const int chunks = (stride * bh + 63) >> 6;                              // 64-slot DMA chunks
        wx0[v] = xmin; wy0[v] = ymin; ws[v] = stride;
        max_stride = stride > max_stride ? stride : max_stride;
        slot0[v] = used;
        used += chunks << 6;
        nch[v + 1] = nch[v] + chunks;
    }
    // ring depth: 3 buffers (DMA two quads ahead) when the windows fit a third of the pool, else 2, else no LDS
    const int cap3 = ((lds_slots - 3 * kZeroSlots) / 3) & ~63, cap2 = ((lds_slots - 2 * kZeroSlots) / 2) & ~63;
    const int nb = used <= cap3 ? 3 : 2;
    const int cap = nb == 3 ? cap3 : cap2;
    const int buf_bytes = kZeroBytes + cap * 16;                               // zero region + the pooled windows
    const bool fits = used <= cap && nch[VT] <= MC * NW && max_stride + 2 <= kZeroSlots;
    TO *const obase = out + (long long)b * C * N;
    const float4 *const fk = featK + (long long)b * VT * nq * HW;                  // this sample's quad planes

    if (fits) {
        for (int i = tid; i < kZeroSlots * nb; i += NT)
            *reinterpret_cast<float4 *>(smem + (i / kZeroSlots) * buf_bytes + (i % kZeroSlots) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- LDS byte offset (inside a buffer) of the nw tap of every view; the sw tap is one row stride further.
        // A sample that is identically zero (!ok: weights 0) reads the zero region at the head of the buffer, which is
        // long enough to hold "one row further" for every admissible stride.
        int a0[VT], ws16[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const bool ok = (valid >> v) & 1u;
            const int s0 = slot0[v] + (ty[v] - wy0[v]) * ws[v] + (tx[v] - wx0[v]);
            a0[v] = ok ? kZeroBytes + s0 * 16 : 0;
            ws16[v] = ws[v] * 16;
        }
        const unsigned voxb = vox * 4u;
        // ---- DMA chunks of this wave: chunk c covers 64 consecutive slots of one view's window
        unsigned g_off[MC];                                              // byte offset inside the sample's quad plane set
        int l_dst[MC];
        int n_c = 0;                                                             // chunks this wave issues per quad
#pragma unroll
        for (int r = 0; r < MC; ++r) {
            const int c = wave + r * NW;
            l_dst[r] = -1;
            g_off[r] = 0;
            if (c < nch[VT]) {
                int v = 0;
#pragma unroll
                for (int u = 1; u < VT; ++u) v += c >= nch[u] ? 1 : 0;
                int sv = ws[0], ox = wx0[0], oy = wy0[0], c0 = nch[0], s0 = slot0[0];
#pragma unroll
                for (int u = 1; u < VT; ++u) if (v == u) { sv = ws[u]; ox = wx0[u]; oy = wy0[u]; c0 = nch[u]; s0 = slot0[u]; }
                const int j = c - c0, slot = (j << 6) + lane;
                const int py = slot / sv, px = slot - py * sv;
                int gx = ox + px, gy = oy + py;                                  // pad column / rows past the window / outside the
                gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);                     // image: clamp -- those slots only meet zero weights
                gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
                g_off[r] = (unsigned)((v * nq) * HW + gy * W + gx) * 16u;
                l_dst[r] = kZeroBytes + (s0 + (j << 6)) * 16;
                ++n_c;
            }
        }
        const unsigned lds_base = (unsigned)(size_t)(lds_void_t *)smem;           // LDS byte address of the ring
        auto ring = [&](int q) { return (nb == 3 ? q % 3 : q & 1) * buf_bytes; };
        auto dma = [&](int q) {
            const float4 *src = fk + (long long)q * HW;
            const int boff = ring(q);
#pragma unroll
            for (int r = 0; r < MC; ++r)
                if (l_dst[r] >= 0) glds16(src, g_off[r], lds_base + (unsigned)uniform(l_dst[r] + boff));
        };
        // ---- channel-quad loop, software-pipelined inside every wave (ring of nb LDS buffers):
        //   iteration q:  issue all 4*V ds_read_b128 of quad q+1         (its buffer was completed by the last barrier)
        //                 cross-view aggregate + store of quad q         (VALU work that covers the LDS latency)
        //                 bilinear fma of quad q+1 -> s                  (first use of the reads)
        //   then          issue the DMA of quad q+nb                    (into the buffer quad q used)
        //                 retire this wave's DMA of quad q+2, barrier
        // The barrier orders LDS traffic only; the counted vmcnt leaves younger DMAs and the output store in flight.
        // Register budget (128 VGPRs at 16 waves/CU, and NO spills: a scratch access is a vector-memory operation and
        // would break the counted vmcnt waits below -- the Makefile fails the build if this kernel uses scratch):
        // one view's taps (4 x b128) are in flight at a time; sub-step u reads view u of quad q+1, aggregates and
        // stores its share of quad q's channels meanwhile, then folds the taps into the next quad's samples.
        f32x4 T[VT][4];                                                          // taps in flight / landed (next quad)
        float s[4][VT];                                                          // samples of the quad being aggregated
        auto read_view = [&](int q, int v) {
            const int base = a0[v] + ring(q), row1 = base + ws16[v];
            T[v][0] = lds_tap(smem, base); T[v][1] = lds_tap(smem, base + 16);
            T[v][2] = lds_tap(smem, row1); T[v][3] = lds_tap(smem, row1 + 16);
        };
        auto fold_view = [&](int v) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i][v] = bilerp(T[v][0].v[i], T[v][1].v[i], T[v][2].v[i], T[v][3].v[i], w00[v], w01[v], w10[v], w11[v]);
                asm volatile("" : "+v"(s[i][v]));      // fold HERE: the optimiser would sink the FMAs to the aggregate and keep T live
            }
        };
        // Stores: one buffer_store_dwordx4 per wave per quad.  A lane quad (4 consecutive z of one column) transposes its
        // 4 channels x 4 voxels in registers, so lane j writes channel j's 16 contiguous bytes; 8 quads of a column make a
        // 128-B run per channel.  (Four dword stores per quad stalled the waves at vector-memory issue: the measured 37 %.)
        // Descriptor rebuilt per quad from wave-uniform values: base = this sample's quad q; per-lane byte offset =
        // channel (lane & 3) * N * 4 + first voxel of the lane quad * 4.
        constexpr unsigned OSZ = sizeof(TO);                                      // fp16 storage: 8 B per lane, 64-B runs per channel
        const unsigned chan_bytes = (unsigned)(N * OSZ);
        const unsigned vox_quad = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(vox * OSZ), 0x00, 0xF, 0xF, false);   // lane 0 of the quad
        const unsigned st_off = vox_quad + (unsigned)(lane & 3) * chan_bytes;
        float res[4];
        auto store_quad = [&](__amdgpu_buffer_rsrc_t rs) {
            quad_transpose(res, lane);
            if constexpr (OSZ == 4) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 d = {__builtin_bit_cast(unsigned, res[0]), __builtin_bit_cast(unsigned, res[1]),
                                 __builtin_bit_cast(unsigned, res[2]), __builtin_bit_cast(unsigned, res[3])};
                __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)st_off, 0, 18);   // nt sc1: the output must not displace the windows in L2
            } else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const __half2 lo = __floats2half2_rn(res[0], res[1]), hi = __floats2half2_rn(res[2], res[3]);   // round to nearest even
                const u32x2 d = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, (int)st_off, 0, 18);
            }
        };
        // one quad.  On entry both halves of quad q's taps are in flight (or landed) in T; the samples s are dead.
        // The taps of quad q+1 are requested half by half as soon as the registers of a half have been folded, so the LDS
        // pipe has 2*VT..4*VT reads of this wave queued during EVERY phase (bilinear FMAs, barrier, DMA issue, aggregate,
        // store) instead of idling while the VALU folds: LDS time and VALU time overlap instead of adding up.
        // The barrier sits between "all waves have folded quad q" (its buffer is free for the DMA of quad q+nb) and the
        // DMA issue; it is a bare s_barrier: the LDS reads in flight across it belong to quad q+1's buffer.
        auto finish_quad = [&](int q) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(obase + (long long)(q * 4) * N, 0, (int)(4u * chan_bytes), 0x00020000);
#pragma unroll
            for (int c = 0; c < 4; ++c) res[c] = aggregate<METHOD, VT>(s[c]);
            store_quad(rs);
        };

        for (int q = 0; q < nb && q < nq; ++q) dma(q);
        wait_vmcnt(0);
        lds_barrier();
        // One quad per iteration.  On entry the taps of quad q are in flight (or landed) in T.  The taps of a view are
        // folded into its sample and the SAME view of quad q+1 is requested at once, so the LDS pipe has this wave's
        // reads queued during the folds, the barrier, the DMA issue, the aggregate and the store -- LDS time and VALU time
        // overlap instead of adding up.  The last view is requested after the store: with all 4*VT taps in flight under
        // the aggregate the kernel does not fit 128 registers.
        // The barrier sits between "every wave has folded quad q" (its buffer is free for the DMA of quad q+nb) and the
        // DMA issue.  It is a bare s_barrier: the reads in flight across it belong to quad q+1's buffer.
        // The loops start at q = -1, an iteration that only requests quad 0: T has a single definition site, inside the
        // loop (a second one in a prologue made the register allocator copy and spill the 4*VT tuples).
#pragma unroll
        for (int v = 0; v < VT; ++v)
#pragma unroll
            for (int t = 0; t < 4; ++t) T[v][t] = f32x4{{0.f, 0.f, 0.f, 0.f}};
        if (nb == 3) {
#pragma nounroll
            for (int q = -1; q < nq; ++q) {
                const int qn = q + 1 < nq ? q + 1 : q;                              // last quad: re-read (unused), no branch
#pragma unroll
                for (int v = 0; v < VT - 1; ++v) {
                    fold_view(v);
                    __builtin_amdgcn_sched_barrier(0);
                    read_view(qn, v);
                    __builtin_amdgcn_sched_barrier(0);
                }
                fold_view(VT - 1);
                __builtin_amdgcn_sched_barrier(0);
                if (q >= 0) {
                    wait_vmcnt(1);                                                 // this wave's DMA of quad q+2 (older than store q-1)
                    bare_barrier();                                                // quad q is folded everywhere; quad q+2 is published
                    if (q + 3 < nq) dma(q + 3);
                    finish_quad(q);
                }
                __builtin_amdgcn_sched_barrier(0);
                read_view(qn, VT - 1);
            }
        } else {
            // 2-deep ring: quad q+1 can only be requested after the barrier that frees quad q-1's buffer, so it is
            // published one barrier later and all reads follow the barrier (rare: windows between a third and half
            // of the pool)
#pragma nounroll
            for (int q = -1; q < nq; ++q) {
                const int qn = q + 1 < nq ? q + 1 : q;
#pragma unroll
                for (int v = 0; v < VT; ++v) fold_view(v);
                if (q >= 0) {
                    wait_vmcnt(1);
                    bare_barrier();                                                // quad q folded everywhere, quad q+1 published
                }
#pragma unroll
                for (int v = 0; v < VT - 1; ++v) read_view(qn, v);
                if (q >= 0) {
                    if (q + 2 < nq) dma(q + 2);
                    finish_quad(q);
                }
                __builtin_amdgcn_sched_barrier(0);
                read_view(qn, VT - 1);
            }
        }
    } else {
        // ---- windows do not fit the LDS pool: sample straight from global memory (clamped taps, zero weights outside)
        int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int x0 = tx[v] < 0 ? 0 : tx[v], y0 = ty[v] < 0 ? 0 : ty[v];
            const int x1 = tx[v] + 1 > W - 1 ? W - 1 : tx[v] + 1, y1 = ty[v] + 1 > H - 1 ? H - 1 : ty[v] + 1;
            const int base = (v * nq) * HW;
            o00[v] = base + y0 * W + x0; o01[v] = base + y0 * W + x1; o10[v] = base + y1 * W + x0; o11[v] = base + y1 * W + x1;
        }
        for (int q = 0; q < nq; ++q) {
            const float4 *src = fk + (long long)q * HW;
            float s[4][VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const float4 a = src[o00[v]], bb = src[o01[v]], c = src[o10[v]], d = src[o11[v]];
                s[0][v] = bilerp(a.x, bb.x, c.x, d.x, w00[v], w01[v], w10[v], w11[v]);
                s[1][v] = bilerp(a.y, bb.y, c.y, d.y, w00[v], w01[v], w10[v], w11[v]);
                s[2][v] = bilerp(a.z, bb.z, c.z, d.z, w00[v], w01[v], w10[v], w11[v]);
                s[3][v] = bilerp(a.w, bb.w, c.w, d.w, w00[v], w01[v], w10[v], w11[v]);
            }
            TO *oq = obase + (long long)(q * 4) * N;
#pragma unroll
            for (int i = 0; i < 4; ++i) (oq + i * N)[vox] = from_f32<TO>(aggregate<METHOD, VT>(s[i]));
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
namespace {
constexpr int kNT = 1024;                             // 2 / 4 views: 1024 voxels per brick (4 x 8 x 32), 128 VGPRs per lane
constexpr int kNT8 = 512;                             // 8 views: 512 voxels per brick (4 x 4 x 32), 256 VGPRs per lane
constexpr int kBYv = kNT / 128;
constexpr int brick_threads(int V) { return V == 8 ? kNT8 : kNT; }

// 16-B LDS slots the ring may use: everything but BrickShared (one block per CU owns all 160 KiB)
int pick_lds_slots() { return (160 * 1024 - 1024) / 16; }

template <int METHOD, int VT, int NT, typename TO>
hipError_t launch_v(const float4 *featK, const float *proj, const float *coords, TO *out, const Problem &p, hipStream_t s)
{
    const int nbx = p.X / kBX, nby = p.Y / (NT / 128), nbz = p.Z / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int slots = pick_lds_slots();
    const size_t lds = (size_t)slots * 16 + sizeof(BrickShared<VT>);
    auto kern = k_fwd_brick<METHOD, VT, NT, TO>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int grid = ((nbx + tiles_x - 1) / tiles_x) * ((nby + tiles_y - 1) / tiles_y) * nbz * 8 * p.B;   // tile work items x 8 XCDs x samples
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, slots, total, make_gate(p, true));
    return hipGetLastError();
}

template <int METHOD, typename TO>
hipError_t launch_m(const float4 *featK, const float *proj, const float *coords, TO *out, const Problem &p, hipStream_t s)
{
    switch (p.V) {
    case 2: return launch_v<METHOD, 2, kNT, TO>(featK, proj, coords, out, p, s);
    case 4: return launch_v<METHOD, 4, kNT, TO>(featK, proj, coords, out, p, s);
    case 8:
        if constexpr (sizeof(TO) == 4) return launch_v<METHOD, 8, kNT8, TO>(featK, proj, coords, out, p, s);   // fp16 store path: over 256 VGPRs
        break;
    }
    return hipErrorNotSupported;
}

template <typename TO>
hipError_t launch_t(const float4 *featK, const float *proj, const float *coords, TO *out, const Problem &p, hipStream_t s)
{
    switch (p.method) {
    case AGG_SOFTMAX: return launch_m<AGG_SOFTMAX, TO>(featK, proj, coords, out, p, s);
    case AGG_SUM: return launch_m<AGG_SUM, TO>(featK, proj, coords, out, p, s);
    case AGG_MEAN: return launch_m<AGG_MEAN, TO>(featK, proj, coords, out, p, s);
    case AGG_MAX: return launch_m<AGG_MAX, TO>(featK, proj, coords, out, p, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace

// ---- geometry gate: one thread per brick projects the brick's 8 corner voxels into every view and sizes the pooled
// windows the brick kernels would need (same arithmetic as their prologue: bbox + 2, odd row stride, 64-slot chunks).
// Voxel centres are affine in the index for every volume the caller builds, so the corners bound the brick's taps.
__global__ void __launch_bounds__(256)
k_brick_gate(const float *__restrict__ proj, const float *__restrict__ coords, int *__restrict__ count, int V, int H, int W, int X,
             int Y, int Z, int by, int nbx, int nby, int nbz, int total, int cap_slots, int max_chunks)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int bps = nbx * nby * nbz;
    const int b = i / bps, r = i % bps;
    const int kz = r % nbz, ky = (r / nbz) % nby, kx = r / (nbz * nby);
    const long long N = (long long)X * Y * Z;
    int used = 0, chunks_all = 0, max_stride = 0;
    for (int v = 0; v < V; ++v) {
        const float *P = proj + ((long long)b * V + v) * 12;
        float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
        bool front = true;
        for (int c = 0; c < 8; ++c) {
            const int vx = kx * kBX + ((c & 1) ? kBX - 1 : 0), vy = ky * by + ((c & 2) ? by - 1 : 0), vz = kz * kBZ + ((c & 4) ? kBZ - 1 : 0);
            const float *Xp = coords + ((long long)b * N + ((long long)vx * Y + vy) * Z + vz) * 3;
            const float a = P[0] * Xp[0] + P[1] * Xp[1] + P[2] * Xp[2] + P[3];
            const float bb = P[4] * Xp[0] + P[5] * Xp[1] + P[6] * Xp[2] + P[7];
            const float z = P[8] * Xp[0] + P[9] * Xp[1] + P[10] * Xp[2] + P[11];
            if (!(z > 0.f)) { front = false; continue; }
            const float ix = (a / z) / (float)H * (float)(W - 1), iy = (bb / z) / (float)W * (float)(H - 1);   // quirk Q1 as in make_taps
            xmin = fminf(xmin, ix); xmax = fmaxf(xmax, ix); ymin = fminf(ymin, iy); ymax = fmaxf(ymax, iy);
        }
        if (!front || xmax < xmin) continue;                                     // behind a camera: the kernels decide per block
        const float x0 = fmaxf(floorf(xmin), -1.f), x1 = fminf(floorf(xmax), (float)(W - 1));
        const float y0 = fmaxf(floorf(ymin), -1.f), y1 = fminf(floorf(ymax), (float)(H - 1));
        if (x1 < x0 || y1 < y0) continue;                                        // wholly outside the image
        const int bw = (int)(x1 - x0) + 2, bh = (int)(y1 - y0) + 2;
        const int stride = bw | 1, chunks = (stride * bh + 63) >> 6;
        used += chunks << 6;
        chunks_all += chunks;
        max_stride = stride > max_stride ? stride : max_stride;
    }
    const bool fits = used <= cap_slots && chunks_all <= max_chunks && max_stride + 2 <= kZeroSlots;
    if (!fits) atomicAdd(count, 1);
}

int brick_count(const Problem &p) { return (p.X / kBX) * (p.Y / (brick_threads(p.V) / 128)) * (p.Z / kBZ) * p.B; }
int brick_fwd_cap_slots() { return ((pick_lds_slots() - 2 * kZeroSlots) / 2) & ~63; }   // the 2-deep ring still stages through LDS

hipError_t launch_brick_gate(const float *proj, const float *coords, int *count, int cap_slots, const Problem &p, hipStream_t s)
{
    const int nt = brick_threads(p.V), by = nt / 128;
    const int nbx = p.X / kBX, nby = p.Y / by, nbz = p.Z / kBZ, total = nbx * nby * nbz * p.B;
    hipLaunchKernelGGL(k_brick_gate, dim3((total + 255) / 256), dim3(256), 0, s, proj, coords, count, p.V, p.H, p.W, p.X, p.Y, p.Z, by,
                       nbx, nby, nbz, total, cap_slots, brick_chunks_per_wave(nt) * (nt / 64));
    return hipGetLastError();
}

bool brick_supported(const Problem &p)
{
    if (p.feat_f16 != p.out_f16) return false;                            // fp32 or fp16 storage throughout; mixed -> gather
    if (p.V != 2 && p.V != 4 && p.V != 8) return false;
    if (p.V == 8 && p.out_f16) return false;
    if (p.C % 4 || p.Z % kBZ || p.X % kBX || p.Y % (brick_threads(p.V) / 128)) return false;
    if ((long long)p.B * p.V * (p.C / 4) * p.H * p.W >= (1ll << 31)) return false;
    if (p.N >= (1ll << 28)) return false;                                 // 32-bit byte offsets inside one quad of the output
    return true;
}

size_t brick_workspace_bytes(const Problem &p)
{
    const size_t n = (size_t)p.B * p.V * p.C * p.H * p.W * sizeof(float);
    return (n + 255) / 256 * 256;
}

hipError_t launch_to_quad_planar(const void *src, void *dst, const Problem &p, hipStream_t s)
{
    if (p.C % 4) return hipErrorNotSupported;
    const int HW = p.H * p.W;
    const dim3 grid((HW + 255) / 256, p.C / 4, p.B * p.V);
    if (p.feat_f16) hipLaunchKernelGGL(k_to_quad_planar<__half>, grid, dim3(256), 0, s, (const __half *)src, (float4 *)dst, p.C, HW, make_gate(p, true));
    else hipLaunchKernelGGL(k_to_quad_planar<float>, grid, dim3(256), 0, s, (const float *)src, (float4 *)dst, p.C, HW, make_gate(p, true));
    return hipGetLastError();
}

hipError_t launch_fwd_brick(const void *featK_, const float *proj, const float *coords, void *out, const Problem &p, hipStream_t s)
{
    if (!brick_supported(p)) return hipErrorNotSupported;
    const float4 *featK = static_cast<const float4 *>(featK_);
    return p.out_f16 ? launch_t<__half>(featK, proj, coords, (__half *)out, p, s) : launch_t<float>(featK, proj, coords, (float *)out, p, s);
}

}  // namespace mvhmr
