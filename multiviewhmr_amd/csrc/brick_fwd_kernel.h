// Fused un-projection forward, "brick" variant: LDS-staged feature windows (gfx950 / CDNA4 only).
//
// Reference semantics: models/aggregation.py:20-87 (projection, depth mask, bilinear grid_sample with zero padding,
// cross-view aggregate); the arithmetic order is pinned in device_common.h.
//
// Mapping.
//   block   = one brick of (4 * NVOX) x (NT / 128) x 32 voxels of one sample, NT threads; a lane owns NVOX voxels (x, x+4),
//             whose tap records (two LDS addresses + 4 weights per view) are computed once and live in registers while the block
//             loops over the C / 4 channel quads;
//   windows = per view, the bounding box of the brick's taps; the views' windows are packed into one LDS buffer.  The staged copy
//             of the features is COLUMN-major quad-planar (B,V,C/4,Wf,Hf,4): a z-long brick seen by an upright camera gives tall
//             narrow windows, so a window column is one contiguous run of the staged copy -- 26 % fewer 128-B line fills than
//             row-major (profiles/r02_fwd_ablations.txt).  In LDS the rows of a column are split by PARITY (MVHMR_FWD_LAY below);
//   ring    = 2 or 3 buffers filled by LDS-DMA (global_load_lds_dwordx4).  Top of quad q: wait for this wave's DMA of quad q,
//             s_barrier (publishes quad q; every wave has folded quad q-1, so that buffer is free), DMA of the next quad;
//   jobs    = (quad q, voxel u): request views 0 and 1, aggregate half of the previous job, fold view 0 / request view 2,
//             fold view 1 / request view 3, aggregate the other half + store, fold views 2 and 3 -- LDS reads, FMAs,
//             transcendentals (issued in runs: device_common.h aggregate2) and the stores are spread over the job;
//   lanes   = fp32 volumes: every lane group the LDS serves in one pass of a ds_read_b128 holds 16 consecutive z of one column
//             (conflict-free with the parity split) and a lane stores its voxel's four channels as four dwords, no transpose;
//             16-bit volumes: the round-3 map -- the 4 channels x 4 voxels of lanes {l, l^4, l^8, l^12} are transposed with DPP
//             row shifts under bank masks, so that a lane QUAD writes 64 contiguous bytes of one channel.
// Two voxels per lane (8 x 8 x 32 bricks) cut the window bytes per voxel by a third.
// Round 4 (profiles/r04_fwd_ablations.txt): the LDS side is no longer what binds (bank conflicts 692 M -> 175 M cycles, LDS busy
// 61 % -> 36 %); what does is the CU's vector-memory pipe (LDS-DMA line fills + volume stores through one TA / TCP) on top of
// ~2.5 ms of VALU issue.
#pragma once
#include "brick_common.h"
#include "kernels.h"

namespace mvhmr {

// Window layout in LDS (MVHMR_FWD_LAY): 0 = plain column-major (slot = column * stride + row); 1 = the rows of a column split by
// PARITY: slot = column * 2 hp + (row & 1) * hp + (row >> 1), window origin row even.  A bilinear footprint {y0, y0 + 1} has one row
// of each parity, so read instruction "even row" / "odd row" of a view fetches, for z-neighbouring lanes (1.45 px apart at the north
// star), the same or the next half-row instead of rows up to 2 apart: the 16 lanes the LDS serves per pass of a ds_read_b128 then
// span ~12 slots instead of ~23 and stop colliding modulo the 16 slots of a pass (scripts/sim_lds5b.py: 4.9 cycles per read
// against 9.7; measured: profiles/r04_fwd_ablations.txt).  hp is rounded to 8 (2 hp = 0 mod 16: the pass residue does not depend on
// the column) when the windows still fit the ring that way.
#ifndef MVHMR_FWD_LAY
#define MVHMR_FWD_LAY 1
#endif
// Lane map (MVHMR_FWD_MAP): 0 = stride-4 transpose map (lane quads 4 z apart, one 16-B store per job); 1 = every lane group of a
// ds_read_b128 pass ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32: MI355X_MICROARCH.md, LDS) holds 16 CONSECUTIVE z of one column, no
// transpose, four dword stores per job (a wave instruction = two 128-B runs).  16-bit volumes keep map 0 (2-B stores do not pay).
#ifndef MVHMR_FWD_MAP
#define MVHMR_FWD_MAP 1
#endif
// Rounding of hp (MVHMR_FWD_HP): 0 never, 1 always to 8, 2 to 8 when the 3-deep ring still fits, else none
#ifndef MVHMR_FWD_HP
#define MVHMR_FWD_HP 2
#endif
#ifndef MVHMR_FWD_MAP16
#define MVHMR_FWD_MAP16 1          // the lane map of 16-bit volumes: 1 = z runs + pair exchange of channel pairs, 0 = round 3's stride-4 transpose
#endif
constexpr int kFwdLay = MVHMR_FWD_LAY, kFwdMapF32 = MVHMR_FWD_MAP, kFwdHp = MVHMR_FWD_HP;
constexpr unsigned kDropOffset = 0xFFFFFFF0u;   // a buffer offset past every num_records this library builds (the range check ignores soffset): that lane's store is dropped
constexpr int kStAux = 18;         // cache policy of the volume stores: nt | sc1 (plain 3.56 ms, sc0 3.55, sc1 3.54, nt 3.44, nt sc1 3.46: r04 ablations)

// map 0: lane = 32 g + 16 h + 4 a + b  ->  column h of the wave's two (x-adjacent) columns, z = 16 g + 4 b + a
// map 1: lane = 32 h + l5; lane quads of l5 -> z quads {0, 16, 20, 4, 24, 8, 12, 28} (+ lane & 3): the LDS pass groups are z runs
template <int MAP>
__device__ __forceinline__ void fwd_lane_voxel(int lane, int &dcol, int &zin)
{
    if constexpr (MAP == 0) {
        dcol = (lane >> 4) & 1;
        zin = ((lane >> 5) << 4) + ((lane & 3) << 2) + ((lane >> 2) & 3);
    } else {
        constexpr unsigned zq = 0u | (4u << 3) | (5u << 6) | (1u << 9) | (6u << 12) | (2u << 15) | (3u << 18) | (7u << 21);
        dcol = lane >> 5;
        zin = (int)(((zq >> (3 * ((lane >> 2) & 7))) & 7u) << 2) + (lane & 3);
    }
}

// DPP move under a bank mask (bank k = lanes 4k..4k+3 of every 16-lane row): masked-off lanes keep `keep`
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_into(float keep, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, x), CTRL, 0xF, BANKS, false));
}

// 4 x 4 transpose across lanes {l, l^4, l^8, l^12} of a 16-lane row: on entry lane 4a+b holds r[i] = value(channel i, z_a);
// on exit it holds r[k] = value(channel a, z_k).  Two butterfly stages of 2 selects + 4 masked DPP moves.
__device__ __forceinline__ void stride4_transpose(float (&r)[4], int lane)
{
    const bool a0 = lane & 4, a1 = lane & 8;
    constexpr int SHL4 = 0x104, SHR4 = 0x114, ROR8 = 0x128;                      // row_shl:4 (from lane+4), row_shr:4 (from lane-4), row_ror:8
    {
        const float x = a0 ? r[0] : r[1], y = a0 ? r[2] : r[3];                  // what the partner (lane ^ 4) takes
        const float n1 = dpp_into<SHL4, 0x5>(r[1], x), n0 = dpp_into<SHR4, 0xA>(r[0], x);
        const float n3 = dpp_into<SHL4, 0x5>(r[3], y), n2 = dpp_into<SHR4, 0xA>(r[2], y);
        r[0] = n0; r[1] = n1; r[2] = n2; r[3] = n3;
    }
    {
        const float x = a1 ? r[0] : r[2], y = a1 ? r[1] : r[3];                  // partner = lane ^ 8
        const float n2 = dpp_into<ROR8, 0x3>(r[2], x), n0 = dpp_into<ROR8, 0xC>(r[0], x);
        const float n3 = dpp_into<ROR8, 0x3>(r[3], y), n1 = dpp_into<ROR8, 0xC>(r[1], y);
        r[0] = n0; r[1] = n1; r[2] = n2; r[3] = n3;
    }
}

// the aggregate of a channel pair; mean over nv <= VT real views: the sum (the absent ones sample zeros) divided by nv -- the reference's
// volume.mean(0) bit for bit, as the gather kernels compute it
template <int METHOD, int VT>
__device__ __forceinline__ void fwd_aggregate2(const float (&sa)[VT], const float (&sb)[VT], float &ra, float &rb, float nvf)
{
    if constexpr (METHOD == AGG_MEAN) {
        ra = __fdiv_rn(aggregate<AGG_SUM, VT>(sa), nvf);
        rb = __fdiv_rn(aggregate<AGG_SUM, VT>(sb), nvf);
    } else {
        aggregate2<METHOD, VT>(sa, sb, ra, rb);
    }
}

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// softmax of one channel pair in the form relative to view 0.  s[0] is view 0's sample (log2(e) * s for PRE), s[v] for v >= 1 the DIFFERENCE to
// it -- folded as such (bilerp_rel: the bilinear sum started from -s[0], one fused operation fewer per view and channel than a sample and a
// subtraction).  out = s0 + sum(e_v d_v) / (1 + sum e_v), e_v = exp(d_v); PRE: e_v = exp2(d_v), the denominator is accumulated times log2(e)
// (v_fmamk), so that its reciprocal already carries the ln 2 of the result: 28 instead of 31 operations per channel against the absolute
// form with its separate scaling.  da + db comes back for the job's overflow test.  Transcendentals in runs (device_common.h aggregate2).
template <int V, bool PRE>
__device__ __forceinline__ void ws_softmax_pair(const float (&sa)[V], const float (&sb)[V], float &ra, float &rb, float &dsum)
{
    float ta[V], tb[V];
#pragma unroll
    for (int v = 1; v < V; ++v) {
        ta[v] = PRE ? sa[v] : sa[v] * kLog2e;
        tb[v] = PRE ? sb[v] : sb[v] * kLog2e;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 1; v < V; ++v) { ta[v] = __builtin_amdgcn_exp2f(ta[v]); tb[v] = __builtin_amdgcn_exp2f(tb[v]); }
    __builtin_amdgcn_sched_barrier(0);
    float da = PRE ? kLog2e : 1.f, db = da;
    float na = ta[1] * sa[1], nb = tb[1] * sb[1];
    if constexpr (PRE) { da = fmaf(ta[1], kLog2e, da); db = fmaf(tb[1], kLog2e, db); }
    else { da += ta[1]; db += tb[1]; }
#pragma unroll
    for (int v = 2; v < V; ++v) {
        if constexpr (PRE) { da = fmaf(ta[v], kLog2e, da); db = fmaf(tb[v], kLog2e, db); }
        else { da += ta[v]; db += tb[v]; }
        na = fmaf(ta[v], sa[v], na);
        nb = fmaf(tb[v], sb[v], nb);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float ia = __builtin_amdgcn_rcpf(da), ib = __builtin_amdgcn_rcpf(db);   // PRE: ln 2 / (1 + sum e_v)
    __builtin_amdgcn_sched_barrier(0);
    ra = fmaf(na, ia, PRE ? sa[0] * kLn2 : sa[0]);
    rb = fmaf(nb, ib, PRE ? sb[0] * kLn2 : sb[0]);
    dsum = da + db;                                                              // both below 2^60 and no NaN: no e_v reached 2^60, so no e_v * d_v overflowed
}

// the max form (any finite samples whose differences to view 0 are finite): the samples are rebuilt from the relative form; aggregate<AGG_SOFTMAX>
// on unscaled samples, on prescaled ones the same with exp2(t - m)
template <int V, bool PRE>
__device__ __forceinline__ float ws_softmax_safe(const float (&r)[V])
{
    float s[V];
    s[0] = r[0];
#pragma unroll
    for (int v = 1; v < V; ++v) s[v] = r[v] + r[0];
    if constexpr (!PRE) {
        return aggregate<AGG_SOFTMAX, V>(s);
    } else {
        float m = s[0];
#pragma unroll
        for (int v = 1; v < V; ++v) m = vmax(m, s[v]);
        float den = 0.f, num = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float e = __builtin_amdgcn_exp2f(s[v] - m);
            den += e;
            num = fmaf(e, s[v], num);
        }
        return num * (__builtin_amdgcn_rcpf(den) * kLn2);
    }
}

// the bilinear sum of a view >= 1 relative to view 0's sample s0 (one rounding per step, the last one on the difference itself)
__device__ __forceinline__ float bilerp_rel(float v00, float v01, float v10, float v11, float w00, float w01, float w10, float w11, float s0)
{
    return __fmaf_rn(v11, w11, __fmaf_rn(v10, w10, __fmaf_rn(v01, w01, __fmaf_rn(v00, w00, -s0))));
}

// s_waitcnt vmcnt(K + n) with an immediate for a wave-uniform n in 0 .. MAXI, tried from the likely end (a wave owns most of its MAXI
// chunk slots): a handful of scalar compares instead of the 21-way switch of wait_vmcnt
template <int K, int I, int MAXI>
__device__ __forceinline__ void wait_vmcnt_ladder(int n)
{
    if constexpr (I <= 0) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory");
    } else {
        if (n >= I) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K + I) : "memory");
        else wait_vmcnt_ladder<K, I - 1, MAXI>(n);
    }
}

template <int VT>
struct FwdShared {
    int bbox[VT][4];               // xmin, ymin, xmax, ymax of the nw taps (valid voxels only)
    float proj[VT][12];
};

// One voxel sampled straight from the staged copy in global memory, channel quads q_begin .. q_end - 1 of the nqv a view holds, channels
// below C only: the path of bricks whose windows do not fit LDS (k_fwd_ws) and of the LAST, partial quad when C % 4 != 0 (k_fwd_tail,
// launched behind every brick kernel: their loops run the C / 4 whole quads).  `unscale` = ln 2 for a copy prescaled by log2(e), else 1.  fk / obase: this sample's quad 0 /
// channel 0.
template <int METHOD, int VT, typename TO>
__device__ __attribute__((noinline)) void fwd_global_voxel(const float4 *fk, TO *obase, const float (*proj)[12], const Coords coords, int b,   // by value: a reference would pin the kernel's copy in scratch
                                                           long long N, unsigned vox, int q_begin, int q_end, int nqv, int C, int H, int W, int nv, float unscale)
{
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;
    const int HW = H * W;
    float c0, c1, c2;
    voxel_xyz(coords, b, N, vox, c0, c1, c2);
    float w00[VT], w01[VT], w10[VT], w11[VT];
    int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        const Taps t = make_taps(proj[v], c0, c1, c2, H, W);
        w00[v] = t.w00; w01[v] = t.w01; w10[v] = t.w10; w11[v] = t.w11;
        const int base = ((v < nv ? v : 0) * nqv) * HW;                         // an absent view reads view 0's pixels (and discards them)
        o00[v] = base + t.x0 * H + t.y0; o01[v] = base + t.x1 * H + t.y0; o10[v] = base + t.x0 * H + t.y1; o11[v] = base + t.x1 * H + t.y1;
    }
    for (int q = q_begin; q < q_end; ++q) {
        const float4 *src = fk + (long long)q * HW;
        float s[4][VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const float4 a = src[o00[v]], bb = src[o01[v]], c = src[o10[v]], d = src[o11[v]];
            s[0][v] = bilerp(a.x, bb.x, c.x, d.x, w00[v], w01[v], w10[v], w11[v]) * unscale;
            s[1][v] = bilerp(a.y, bb.y, c.y, d.y, w00[v], w01[v], w10[v], w11[v]) * unscale;
            s[2][v] = bilerp(a.z, bb.z, c.z, d.z, w00[v], w01[v], w10[v], w11[v]) * unscale;
            s[3][v] = bilerp(a.w, bb.w, c.w, d.w, w00[v], w01[v], w10[v], w11[v]) * unscale;
            if (v >= nv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i][v] = kAbsentReads ? kAbsentSample : 0.f;
            }
        }
        TO *oq = obase + (long long)(q * 4) * N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r;
            if constexpr (METHOD == AGG_MEAN) r = __fdiv_rn(aggregate<AGG_SUM, VT>(s[i]), (float)nv);
            else r = aggregate<METHOD, VT>(s[i]);
            if (q * 4 + i < C) (oq + i * N)[vox] = from_f32<TO>(r);
        }
    }
}

// pool geometry shared by the kernel, the gate and the host: `slots` 16-B slots hold nb buffers of kZeroSlots + cap slots
__host__ __device__ inline int fwd_cap3(int slots) { return ((slots - 3 * kZeroSlots) / 3) & ~63; }
__host__ __device__ inline int fwd_cap2(int slots) { return ((slots - 2 * kZeroSlots) / 2) & ~63; }

template <int METHOD, int VT, int NT, typename TO, int NVOX>
__global__ void __launch_bounds__(NT)
k_fwd_brick(const float4 *__restrict__ featK, const float *__restrict__ proj, const Coords coords,
            TO *__restrict__ out, int C, int H, int W, int X, int Y, int Z, int nby, int nbz, int bricks_per_sample,
            int lds_slots, int total_blocks, int nv, int ksplit, Gate gate)
{
    // ksplit > 1 (launches of fewer bricks than CUs: single-sample inference): the channel quads of a brick are divided among ksplit blocks
    // nv <= VT views are real (3 views run the 4-view kernel): the others have no camera, no window and no part in the aggregate --
    // their samples read kAbsentSample from a slot of the zero region (softmax weight exp(-FLT_MAX - m) = 0, never the maximum) or
    // plain zeros (sum; mean, which is rescaled by VT / nv)
    if (gated_off(gate)) return;
    constexpr int BY = NT / 128, NW = NT / 64, BXK = kBX * NVOX;
    constexpr int MAP = sizeof(TO) == 4 ? kFwdMapF32 : MVHMR_FWD_MAP16, LAY = kFwdLay;
    constexpr int SPJ = MAP == 1 ? (sizeof(TO) == 4 ? 4 : 2) : 1;                // store instructions per job
    constexpr int MC = brick_chunks_per_wave(NT);                                 // DMA chunks a wave may own per quad
    extern __shared__ __align__(16) unsigned char smem[];
    FwdShared<VT> *sh = reinterpret_cast<FwdShared<VT> *>(smem + lds_slots * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform((int)(tid >> 6));

    // XCD-aware order (speed only).  Blocks i, i+8, ... share an XCD under round-robin dispatch.  All eight XCDs work on the
    // same sample at a time, each on a compact tile of brick columns (all z): neighbouring windows meet in one L2 and the live
    // feature planes stay in the Infinity Cache.
    const int nbx = bricks_per_sample / (nby * nbz);
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int tw = (nbx + tiles_x - 1) / tiles_x, th = (nby + tiles_y - 1) / tiles_y;
    const int share = tw * th * nbz;
    const int grid0 = (int)gridDim.x / ksplit, part = (int)blockIdx.x / grid0, bid = (int)blockIdx.x - part * grid0;
    const int xcd = bid & 7, j = bid >> 3;
    const int b = j / share, r = j % share;
    const int kz = r % nbz, cy = (r / nbz) % th, cx = r / (nbz * th);
    const int kx = (xcd % tiles_x) * tw + cx, ky = (xcd / tiles_x) * th + cy;
    if (kx >= nbx || ky >= nby || b * bricks_per_sample >= total_blocks) return;
    const long long N = (long long)X * Y * Z;
    const int HW = H * W, nqv = (C + 3) >> 2, nq = (C >> 2) / ksplit, q0 = part * nq;   // nqv: quads per view (stride); nq: this block's WHOLE quads, from q0

    if (tid < VT * 12) sh->proj[tid / 12][tid % 12] = tid < nv * 12 ? proj[((long long)b * nv) * 12 + tid] : 0.f;
    if (tid < VT) { sh->bbox[tid][0] = 1 << 30; sh->bbox[tid][1] = 1 << 30; sh->bbox[tid][2] = -(1 << 30); sh->bbox[tid][3] = -(1 << 30); }
    __syncthreads();

    // ---- this lane's voxels and their tap records (once per brick)
    int dcol, zin;
    fwd_lane_voxel<MAP>(lane, dcol, zin);
    const int col = wave * 2 + dcol;
    // Volumes need not divide into bricks (r04): a lane whose voxel lies outside the volume works on the clamped edge voxel's centre,
    // takes no part in the windows (its samples read the zero region) and its stores are dropped by the buffer range check.
    const int vy_r = ky * BY + (col >> 2), vz_r = kz * kBZ + zin;
    const bool in_yz = vy_r < Y && vz_r < Z;
    const int vy = vy_r < Y ? vy_r : Y - 1, vz = vz_r < Z ? vz_r : Z - 1;
    unsigned vox[NVOX];
    bool inside[NVOX];
    float w00[NVOX][VT], w01[NVOX][VT], w10[NVOX][VT], w11[NVOX][VT];
    int tx[NVOX][VT], ty[NVOX][VT];
    unsigned valid = 0;
    {
        int bxmin[VT], bymin[VT], bxmax[VT], bymax[VT];
        const int big = 1 << 30;
#pragma unroll
        for (int v = 0; v < VT; ++v) { bxmin[v] = big; bymin[v] = big; bxmax[v] = -big; bymax[v] = -big; }
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            const int vx_r = kx * BXK + (col & 3) + kBX * u;
            inside[u] = in_yz && vx_r < X;
            const int vx = vx_r < X ? vx_r : X - 1;
            vox[u] = (unsigned)(((long long)vx * Y + vy) * Z + vz);             // N < 2^28 (brick_fwd_supported)
            float c0, c1, c2;
            voxel_xyz(coords, b, N, vox[u], c0, c1, c2);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const Taps t = make_taps(sh->proj[v], c0, c1, c2, H, W);
                w00[u][v] = t.w00; w01[u][v] = t.w01; w10[u][v] = t.w10; w11[u][v] = t.w11;
                tx[u][v] = t.rx0; ty[u][v] = t.ry0;
                if (t.any && inside[u] && v < nv) {
                    valid |= 1u << (u * VT + v);
                    bxmin[v] = t.rx0 < bxmin[v] ? t.rx0 : bxmin[v]; bxmax[v] = t.rx0 > bxmax[v] ? t.rx0 : bxmax[v];
                    bymin[v] = t.ry0 < bymin[v] ? t.ry0 : bymin[v]; bymax[v] = t.ry0 > bymax[v] ? t.ry0 : bymax[v];
                }
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

    // ---- window per view (block-uniform): origin, column stride (odd) in slots, first slot; views packed back to back
    // LAY 1: origin row even, hp half-rows per parity, column stride 2 hp; hp rounded to 8 by policy kFwdHp
    int wx0[VT], wy0[VT], ws[VT], whp[VT], nch[VT + 1], slot0[VT];
    const int cap3 = fwd_cap3(lds_slots), cap2 = fwd_cap2(lds_slots);
    int used = 0, max_stride = 0;
    auto size_windows = [&](bool round8) __attribute__((always_inline)) {
        nch[0] = 0; used = 0; max_stride = 0;
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            const int xmin = uniform(sh->bbox[v][0]), ymin = uniform(sh->bbox[v][1]);
            const int xmax = uniform(sh->bbox[v][2]), ymax = uniform(sh->bbox[v][3]);
            int bw = 0, stride = 1, hp = 0, y0w = ymin;
            if constexpr (LAY == 0) {
                int bh = 0;
                if (xmax >= xmin) { bw = xmax - xmin + 2; bh = ymax - ymin + 2; }   // taps reach x0+1, y0+1
                stride = bh | 1;
            } else {
                y0w = ymin & ~1;
                if (xmax >= xmin) { bw = xmax - xmin + 2; hp = (ymax + 3 - y0w) >> 1; }   // rows y0w .. ymax + 1
                if (round8) hp = (hp + 7) & ~7;
                stride = 2 * hp;
            }
            const int chunks = (stride * bw + 63) >> 6;                          // 64-slot DMA chunks
            wx0[v] = xmin; wy0[v] = y0w; ws[v] = stride; whp[v] = hp;
            max_stride = stride > max_stride ? stride : max_stride;
            slot0[v] = used;
            used += chunks << 6;
            nch[v + 1] = nch[v] + chunks;
        }
    };
    if constexpr (LAY == 0 || kFwdHp == 0) {
        size_windows(false);
    } else if constexpr (kFwdHp == 1) {
        size_windows(true);
        if (!(used <= cap2 && nch[VT] <= MC * NW && max_stride + 2 <= kZeroSlots)) size_windows(false);
    } else {
        size_windows(true);
        if (!(used <= cap3 && nch[VT] <= MC * NW && max_stride + 2 <= kZeroSlots)) size_windows(false);
    }
    // ring depth: 3 buffers (a DMA has between one and two iterations to land) when the windows fit a third of the pool, else 2
    const int nb = used <= cap3 ? 3 : 2;
    const int cap = nb == 3 ? cap3 : cap2;
    const int buf_bytes = kZeroBytes + cap * 16;
    const bool fits = used <= cap && nch[VT] <= MC * NW && max_stride + 2 <= kZeroSlots;
    TO *const obase = out + (long long)b * C * N + (long long)(q0 * 4) * N;
    const float4 *const fk = featK + (long long)b * nv * nqv * HW + (long long)q0 * HW;
    constexpr bool kAbsentReads = METHOD == AGG_SOFTMAX || METHOD == AGG_MAX;    // absent views sample kAbsentSample (else zeros)
    const float mean_fix = (float)nv;                                             // the mean's divisor: the real views

    if (fits) {
        for (int i = tid; i < kZeroSlots * nb; i += NT) {
            const float z = (kAbsentReads && nv < VT && i % kZeroSlots == kAbsentSlot) ? kAbsentSample : 0.f;
            *reinterpret_cast<float4 *>(smem + (i / kZeroSlots) * buf_bytes + (i % kZeroSlots) * 16) = make_float4(z, z, z, z);
        }
        // ---- LDS byte offsets (inside a buffer) of the taps in column x0; column x0 + 1 is one stride further.  A sample that is
        // identically zero reads the zero region at the head of the buffer (long enough for "one stride further").
        // LAY 0: a0 = the nw tap, sw 16 B further.  LAY 1: a0 = the EVEN row of the footprint, a1 = the odd row, and the weights are
        // kept in that order (even row x0, even row x0+1, odd row x0, odd row x0+1): for an odd y0 the sum runs sw, se, nw, ne
        // instead of ATen's nw, ne, sw, se -- one rounding order among equals (<= 1 ulp of the sample)
        int a0[NVOX][VT], a1[LAY ? NVOX : 1][VT], ws16[VT];
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            ws16[v] = ws[v] * 16;
#pragma unroll
            for (int u = 0; u < NVOX; ++u) {
                const bool ok = (valid >> (u * VT + v)) & 1u;
                if constexpr (LAY == 0) {
                    const int s0 = slot0[v] + (tx[u][v] - wx0[v]) * ws[v] + (ty[u][v] - wy0[v]);
                    a0[u][v] = ok ? kZeroBytes + s0 * 16 : 0;
                } else {
                    const int yr = ty[u][v] - wy0[v];
                    const int sc = slot0[v] + (tx[u][v] - wx0[v]) * ws[v];
                    a0[u][v] = ok ? kZeroBytes + (sc + ((yr + 1) >> 1)) * 16 : 0;
                    a1[u][v] = ok ? kZeroBytes + (sc + whp[v] + (yr >> 1)) * 16 : 0;
                    if (yr & 1) {
                        const float t0 = w00[u][v], t1 = w01[u][v];
                        w00[u][v] = w10[u][v]; w01[u][v] = w11[u][v]; w10[u][v] = t0; w11[u][v] = t1;
                    }
                }
                if (kAbsentReads && v >= nv) {                                   // wave-uniform: the absent view's one "tap"
                    a0[u][v] = kAbsentSlot * 16;
                    if constexpr (LAY != 0) a1[u][v] = kAbsentSlot * 16;
                    w00[u][v] = 1.f; w01[u][v] = 0.f; w10[u][v] = 0.f; w11[u][v] = 0.f;
                }
            }
            if (kAbsentReads && v >= nv) ws16[v] = 16;                           // "one stride further": the zero slot next to it
        }
        // ---- DMA chunks of this wave: chunk c covers 64 consecutive slots of one view's window; the wave owns chunks wave,
        // wave + NW, ...: rr < n_c of its MC slots
        unsigned g_off[MC];
        int l_dst[MC];
        int n_c = 0;
#pragma unroll
        for (int rr = 0; rr < MC; ++rr) {
            const int c = wave + rr * NW;
            l_dst[rr] = 0;
            g_off[rr] = 0;
            if (c < nch[VT]) {
                int v = 0;
#pragma unroll
                for (int uu = 1; uu < VT; ++uu) v += c >= nch[uu] ? 1 : 0;
                int sv = ws[0], ox = wx0[0], oy = wy0[0], c0 = nch[0], s0 = slot0[0], hv = whp[0];
#pragma unroll
                for (int uu = 1; uu < VT; ++uu) if (v == uu) { sv = ws[uu]; ox = wx0[uu]; oy = wy0[uu]; c0 = nch[uu]; s0 = slot0[uu]; hv = whp[uu]; }
                const int jj = c - c0, slot = (jj << 6) + lane;
                const int px = slot / sv;
                int py = slot - px * sv;
                if constexpr (LAY == 1) py = py >= hv ? 2 * (py - hv) + 1 : 2 * py;   // slot inside the column -> row
                int gx = ox + px, gy = oy + py;                                  // pad rows / columns past the window / outside the
                gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);                     // image: clamp -- those slots only meet zero weights
                gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
                g_off[rr] = (unsigned)((v * nqv) * HW + gx * H + gy) * 16u;
                l_dst[rr] = uniform(kZeroBytes + (s0 + (jj << 6)) * 16) + (int)(unsigned)(size_t)(lds_void_t *)smem;
                ++n_c;
            }
        }
        // What changes per quad is carried in scalar registers: the ring offsets of quads q, q+1, q+2 (rotated, no q % 3), the plane
        // of the staged copy the next request reads (one pointer add), this quad's and the previous quad's store descriptors.
        int r0 = 0, r1 = buf_bytes, r2 = 2 * buf_bytes;
        const float4 *src_n = fk;                                                // plane of the next quad to request
        auto dma = [&](int roff) __attribute__((always_inline)) {
#pragma unroll
            for (int rr = 0; rr < MC; ++rr)
                if (rr < n_c) glds16_m0(src_n, g_off[rr], (unsigned)(l_dst[rr] + roff));
            src_n += HW;
        };

        // ---- stores.  Map 1: the lane's own voxel, channel i at soffset i * chan_bytes (a wave instruction = two 128-B runs of one
        // channel plane; the TA coalesces it in 4 cycles: TA_BUFFER_COALESCED_WRITE_CYCLES).  Map 0: lane 4a+b (+16h+32g) writes
        // channel a, z = 16g + 4b .. 4b+3 of its column after the stride-4 transpose.
        constexpr unsigned OSZ = sizeof(TO);
        const unsigned chan_bytes = (unsigned)(N * OSZ);
        unsigned st_off[NVOX];
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            if constexpr (MAP == 1 && OSZ == 2) {
                // 16-bit volume: lanes 2m / 2m+1 (z, z+1 of one column) exchange channel pairs; the even lane writes (z, z+1) of
                // channels 0 / 1 as one dword each, the odd lane those of channels 2 / 3: 64-B runs per channel and column
                // (Z even: a pair is inside or outside the volume as a whole)
                st_off[u] = inside[u] ? (vox[u] - (unsigned)(lane & 1)) * OSZ + (unsigned)(lane & 1) * 2u * chan_bytes : kDropOffset;
            } else if constexpr (MAP == 1) {
                st_off[u] = inside[u] ? vox[u] * OSZ : kDropOffset;               // beyond num_records (<= 2^32 - 16): the store is dropped
            } else {
                static_assert(MAP == 1, "the stride-4 transpose map writes four z per lane: whole bricks only");
                const int z0 = ((lane >> 5) << 4) + ((lane & 3) << 2);
                st_off[u] = (vox[u] - (unsigned)zin + (unsigned)z0) * OSZ + (unsigned)((lane >> 2) & 3) * chan_bytes;
            }
        }
        auto make_rs = [&](TO *base) __attribute__((always_inline)) {
            return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(4u * chan_bytes), 0x00020000);
        };
        auto store_quad = [&](const __amdgpu_buffer_rsrc_t rs, int u, float (&res)[4]) __attribute__((always_inline)) {
            if constexpr (MAP == 1 && OSZ == 2) {
                const bool odd = lane & 1;
                const float s0 = odd ? res[0] : res[2], s1 = odd ? res[1] : res[3];     // what the partner (lane ^ 1) takes
                const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, false));
                const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, false));
                // even lane: (own channel i at z, partner's channel i at z+1); odd lane: (partner's channel 2+i at z-1, own at z)
                const unsigned d0 = odd ? pack2<TO>(g0, res[2]) : pack2<TO>(res[0], g0);
                const unsigned d1 = odd ? pack2<TO>(g1, res[3]) : pack2<TO>(res[1], g1);
                __builtin_amdgcn_raw_buffer_store_b32(d0, rs, (int)st_off[u], 0, kStAux);
                __builtin_amdgcn_raw_buffer_store_b32(d1, rs, (int)st_off[u], (int)chan_bytes, kStAux);
            } else if constexpr (MAP == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[i]), rs, (int)st_off[u], (int)(i * chan_bytes), kStAux);
            } else {
                stride4_transpose(res, lane);
                if constexpr (OSZ == 4) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 d = {__builtin_bit_cast(unsigned, res[0]), __builtin_bit_cast(unsigned, res[1]),
                                     __builtin_bit_cast(unsigned, res[2]), __builtin_bit_cast(unsigned, res[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)st_off[u], 0, kStAux);
                } else {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 d = {pack2<TO>(res[0], res[1]), pack2<TO>(res[2], res[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(d, rs, (int)st_off[u], 0, kStAux);
                }
            }
        };

        // ---- taps: two register sets of 4 x b128 (LAY 0: nw, ne, sw, se; LAY 1: even row x0 / x0+1, odd row x0 / x0+1), used
        // alternately by consecutive views
        f32x4 T[2][4];
        // (the ring offset and the column strides as VGPR operands of the address adds -- full-rate instead of half-rate v_add_u32 --
        // were measured at nothing: profiles/r04_fwd_ablations.txt section L)
        auto read_view = [&](int roff, int u, int v, int set) __attribute__((always_inline)) {
            const int base = a0[u][v] + roff, far = base + ws16[v];
            if constexpr (LAY == 0) {
                T[set][0] = lds_tap(smem, base); T[set][2] = lds_tap(smem, base + 16);
                T[set][1] = lds_tap(smem, far); T[set][3] = lds_tap(smem, far + 16);
            } else {
                const int base1 = a1[u][v] + roff, far1 = base1 + ws16[v];
                T[set][0] = lds_tap(smem, base); T[set][2] = lds_tap(smem, base1);
                T[set][1] = lds_tap(smem, far); T[set][3] = lds_tap(smem, far1);
            }
        };
        float sq[4][VT], sp[4][VT], res[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { T[0][t] = f32x4{{0.f, 0.f, 0.f, 0.f}}; T[1][t] = f32x4{{0.f, 0.f, 0.f, 0.f}}; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            res[i] = 0.f;
#pragma unroll
            for (int v = 0; v < VT; ++v) { sq[i][v] = 0.f; sp[i][v] = 0.f; }
        }
        dma(0);                                                                  // quad 0 -> buffer 0
        if (nb == 3 && nq > 1) dma(r1);                                          // quad 1 -> buffer 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the zero regions are written
        __amdgpu_buffer_rsrc_t rs_cur = make_rs(obase), rs_prev = rs_cur;
        TO *oq_cur = obase;
        const long long qstride = 4 * N;                                         // elements between the channel planes of consecutive quads

        // One quad.  Top: counted wait for this wave's DMA of quad q, s_barrier (publishes quad q; every wave has folded quad q-1, so
        // that buffer is free); two buffers: request quad q+1 here.  Jobs (quad q, voxel u): request views 0 and 1, aggregate half of
        // the PREVIOUS job, fold view 0 / request view 2, fold view 1 / request view 3 (three buffers: behind it, in the first job,
        // the LDS-DMA of quad q+2 -- ONE site per quad), aggregate the other half + store, fold views 2 and 3.
        // PAR: which of sq / sp receives the samples of job u = 0 (alternates per quad when NVOX is odd).
        // The counted wait: the DMA of quad q must have landed; younger than it are the stores of NVOX jobs (two buffers) or of
        // 2 NVOX - 1 jobs plus the n_c pieces of quad q+1 (three buffers).  The first and the last iterations lack part of that order.
        // the DMA site: first job, behind the fold of the view whose head carries the stores -- a request burst in FRONT of the
        // stores holds them up in the TA / TCP (site behind fold 1: 3.41 ms, behind fold 2 or 3: 3.34; pieces or stores spread over
        // the job: 3.39-3.62: profiles/r04_fwd_ablations.txt)
        constexpr int DMA_U = 0, DMA_V = (VT + 1) / 2 < VT ? (VT + 1) / 2 : VT - 1;
        auto quad_iter = [&](int q, auto par_tag) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
            {
                // three buffers: store groups behind the DMA site in its own quad + everything of the next quad
                constexpr int K2 = NVOX * SPJ, K3 = (NVOX - DMA_U - (DMA_V >= (VT + 1) / 2 ? 1 : 0)) * SPJ + NVOX * SPJ;
                static_assert(K3 + MC <= 63, "vmcnt is a 6-bit field");
                if (q < 2 || q + 1 >= nq) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (nb == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K2) : "memory");
                else wait_vmcnt_ladder<K3, MC, MC>(n_c);
            }
            bare_barrier();
            if (nb == 2 && q + 1 < nq) dma(r1);
            const int r0v = r0;                                                  // this quad's ring offset for the tap addresses
#pragma unroll
            for (int u = 0; u < NVOX; ++u) {
                // this job's samples go to sq / sp alternately; the previous job's are aggregated in two halves between the folds
                // (the live samples stay 16 registers: rows of `prev` die as columns of `cur` are born)
                auto &cur = ((u + PAR) & 1) ? sp : sq;
                auto &prev = ((u + PAR) & 1) ? sq : sp;
                read_view(r0v, u, 0, 0);
                if constexpr (VT > 1) read_view(r0v, u, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                fwd_aggregate2<METHOD, VT>(prev[0], prev[1], res[0], res[1], mean_fix);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < VT; ++v) {
                    if (v == (VT + 1) / 2) {
                        fwd_aggregate2<METHOD, VT>(prev[2], prev[3], res[2], res[3], mean_fix);
                        if (u > 0) store_quad(rs_cur, u - 1, res);
                        else if (q > 0) store_quad(rs_prev, NVOX - 1, res);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        cur[i][v] = bilerp(T[v & 1][0].v[i], T[v & 1][1].v[i], T[v & 1][2].v[i], T[v & 1][3].v[i], w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                        asm volatile("" : "+v"(cur[i][v]));                       // fold HERE: keeps the tap registers short-lived
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (v + 2 < VT) read_view(r0v, u, v + 2, v & 1);
                    if (u == DMA_U && v == DMA_V && nb == 3 && q + 2 < nq) dma(r2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // next quad: rotate the ring offsets, advance the store descriptors
            const int t0 = r0;
            r0 = r1; r1 = nb == 3 ? r2 : t0; r2 = t0;
            rs_prev = rs_cur;
            oq_cur += qstride;
            rs_cur = make_rs(oq_cur);
        };
        if constexpr (NVOX & 1) {
            for (int q = 0; q < nq; q += 2) {
                quad_iter(q, std::integral_constant<int, 0>{});
                if (q + 1 < nq) quad_iter(q + 1, std::integral_constant<int, 1>{});
            }
        } else {
            for (int q = 0; q < nq; ++q) quad_iter(q, std::integral_constant<int, 0>{});
        }
        // the last job: (nq - 1, NVOX - 1); rs_prev is the descriptor of quad nq - 1 by now
        const bool last_in_sp = ((NVOX & 1) ? nq - 1 : NVOX - 1) & 1;
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            if (last_in_sp) fwd_aggregate2<METHOD, VT>(sp[c], sp[c + 1], res[c], res[c + 1], mean_fix);
            else fwd_aggregate2<METHOD, VT>(sq[c], sq[c + 1], res[c], res[c + 1], mean_fix);
        }
        store_quad(rs_prev, NVOX - 1, res);
    } else {
        // ---- windows do not fit the LDS pool: sample straight from global memory (clamped taps, zero weights outside)
#pragma unroll
        for (int u = 0; u < NVOX; ++u) {
            int o00[VT], o01[VT], o10[VT], o11[VT];
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                const int x0 = tx[u][v] < 0 ? 0 : tx[u][v], y0 = ty[u][v] < 0 ? 0 : ty[u][v];
                const int x1 = tx[u][v] + 1 > W - 1 ? W - 1 : tx[u][v] + 1, y1 = ty[u][v] + 1 > H - 1 ? H - 1 : ty[u][v] + 1;
                const int base = ((v < nv ? v : 0) * nqv) * HW;                 // an absent view reads view 0's pixels (and discards them)
                o00[v] = base + x0 * H + y0; o01[v] = base + x1 * H + y0; o10[v] = base + x0 * H + y1; o11[v] = base + x1 * H + y1;
            }
            for (int q = 0; q < nq; ++q) {
                const float4 *src = fk + (long long)q * HW;
                float s[4][VT];
#pragma unroll
                for (int v = 0; v < VT; ++v) {
                    const float4 a = src[o00[v]], bb = src[o01[v]], c = src[o10[v]], d = src[o11[v]];
                    s[0][v] = bilerp(a.x, bb.x, c.x, d.x, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    s[1][v] = bilerp(a.y, bb.y, c.y, d.y, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    s[2][v] = bilerp(a.z, bb.z, c.z, d.z, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    s[3][v] = bilerp(a.w, bb.w, c.w, d.w, w00[u][v], w01[u][v], w10[u][v], w11[u][v]);
                    if (v >= nv) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) s[i][v] = kAbsentReads ? kAbsentSample : 0.f;
                    }
                }
                TO *oq = obase + (long long)(q * 4) * N;
                if (inside[u]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float r;
                        if constexpr (METHOD == AGG_MEAN) r = __fdiv_rn(aggregate<AGG_SUM, VT>(s[i]), mean_fix);
                        else r = aggregate<METHOD, VT>(s[i]);
                        (oq + i * N)[vox[u]] = from_f32<TO>(r);
                    }
                }
            }
        }
    }
}

// Few bricks per CU (single-sample inference: 64^3 = 128 bricks on 256 CUs; batch 3: a second round that is half empty): every brick's
// channel quads are divided among 2 or 4 blocks when that shortens the launch.  Cost model per round of 256 blocks: ~20 us of prologue
// (tap records, windows, first DMA) + ~3 us per quad (north star: 0.21 ms per round of 64 quads); measured batch 1: 0.181 -> 0.128 ms.
inline int brick_fwd_ksplit(int bricks, int quads)
{
    int best = 1;
    double best_t = 1e30;
    for (int ks = 1; ks <= 4; ks *= 2) {
        if (quads % ks || (ks > 1 && quads / ks < 8)) break;
        const double t = (double)((bricks * ks + 255) / 256) * (20.0 + 3.0 * quads / ks);
        if (t < best_t * 0.97) { best_t = t; best = ks; }                         // a split has to pay at least 3 %
    }
    return best;
}

// ---- launch of one instantiation (shared by the per-method translation units)
int fwd_lds_slots();                                  // 16-B LDS slots of the ring (one block per CU owns all 160 KiB)

template <int METHOD, int VT, int NT, typename TO, int NVOX>
hipError_t launch_fwd_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s)
{
    const int nbx = (p.X + kBX * NVOX - 1) / (kBX * NVOX), nby = (p.Y + NT / 128 - 1) / (NT / 128), nbz = (p.Z + kBZ - 1) / kBZ;
    const int bps = nbx * nby * nbz, total = bps * p.B;
    const int slots = fwd_lds_slots();
    const size_t lds = (size_t)slots * 16 + sizeof(FwdShared<VT>);
    auto kern = k_fwd_brick<METHOD, VT, NT, TO, NVOX>;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    const int tiles_x = nbx >= nby ? 4 : 2, tiles_y = 8 / tiles_x;
    const int grid = ((nbx + tiles_x - 1) / tiles_x) * ((nby + tiles_y - 1) / tiles_y) * nbz * 8 * p.B;   // tile work items x 8 XCDs x samples
    const int ks = brick_fwd_ksplit(total, p.C / 4);
    hipLaunchKernelGGL(kern, dim3(grid * ks), dim3(NT), lds, s, featK, proj, coords, out, p.C, p.H, p.W, p.X, p.Y, p.Z, nby, nbz, bps, slots, total, p.V,
                       ks, make_gate(p, true));
    return hipGetLastError();
}

// 8 views in two groups of four, 1024 threads (brick_fwd_groups.h)
bool brick_fwd_grouped(const Problem &p);
template <int METHOD, int VT, typename TO>
hipError_t launch_fwd_groups_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s);

// 3 / 4 views, chip-filling launches: the wave-specialised kernel (brick_fwd_ws.h); PRE: the staged copy is multiplied by log2(e)
bool brick_fwd_ws_shape(const Problem &p);
template <int METHOD, bool PRE, typename TO>
hipError_t launch_fwd_ws_instance(const float4 *featK, const float *proj, const Coords &coords, TO *out, const Problem &p, hipStream_t s);
template <int METHOD, bool PRE>
hipError_t launch_fwd_ws_storage(const float4 *featK, const float *proj, const Coords &coords, void *out, const Problem &p, hipStream_t s)
{
    return p.out_f16    ? launch_fwd_ws_instance<METHOD, PRE, __half>(featK, proj, coords, (__half *)out, p, s)
           : p.out_bf16 ? launch_fwd_ws_instance<METHOD, PRE, bf16_t>(featK, proj, coords, (bf16_t *)out, p, s)
                        : launch_fwd_ws_instance<METHOD, PRE, float>(featK, proj, coords, (float *)out, p, s);
}

// one aggregation method: views x storage type x voxels per lane
template <int METHOD>
hipError_t launch_fwd_method(const void *featK_, const float *proj, const Coords &coords, void *out, const Problem &p, int nvox, hipStream_t s)
{
    const float4 *featK = static_cast<const float4 *>(featK_);
    if (brick_fwd_ws_shape(p)) {
        if constexpr (METHOD == AGG_SOFTMAX) {
            if (p.feat_log2e) return launch_fwd_ws_storage<METHOD, true>(featK, proj, coords, out, p, s);
        }
        if (p.feat_log2e) return hipErrorNotSupported;
        return launch_fwd_ws_storage<METHOD, false>(featK, proj, coords, out, p, s);
    }
    if (p.feat_log2e) return hipErrorNotSupported;                               // only the wave-specialised softmax reads a prescaled copy
#define MVHMR_FWD_CASE(NVIEWS, NTHR, NV)                                                                                                 \
    if (brick_view_slots(p.V) == NVIEWS && nvox == NV)                                                                                                   \
        return p.out_f16    ? launch_fwd_instance<METHOD, NVIEWS, NTHR, __half, NV>(featK, proj, coords, (__half *)out, p, s)            \
               : p.out_bf16 ? launch_fwd_instance<METHOD, NVIEWS, NTHR, bf16_t, NV>(featK, proj, coords, (bf16_t *)out, p, s)            \
                            : launch_fwd_instance<METHOD, NVIEWS, NTHR, float, NV>(featK, proj, coords, (float *)out, p, s)
    if (brick_fwd_grouped(p))
        return p.out_f16    ? launch_fwd_groups_instance<METHOD, 8, __half>(featK, proj, coords, (__half *)out, p, s)
               : p.out_bf16 ? launch_fwd_groups_instance<METHOD, 8, bf16_t>(featK, proj, coords, (bf16_t *)out, p, s)
                            : launch_fwd_groups_instance<METHOD, 8, float>(featK, proj, coords, (float *)out, p, s);
    MVHMR_FWD_CASE(2, 1024, 1); MVHMR_FWD_CASE(2, 1024, 2);
    MVHMR_FWD_CASE(4, 1024, 1); MVHMR_FWD_CASE(4, 1024, 2);
    MVHMR_FWD_CASE(8, 512, 1);  MVHMR_FWD_CASE(8, 512, 2);
#undef MVHMR_FWD_CASE
    return hipErrorNotSupported;
}

}  // namespace mvhmr
