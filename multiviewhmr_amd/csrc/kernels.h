// Host-visible launchers of the HIP kernels (internal to the library; the public ABI is include/mvhmr_unproject.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "gate.h"

namespace mvhmr {

struct Problem {
    int B, V, C, H, W;        // features (B,V,C,H,W)
    int X, Y, Z;              // volume
    long long N;              // X*Y*Z
    int C4;                   // channels rounded up to a multiple of 4 (channels-last row length)
    int method;               // AGG_*
    int feat_f16, out_f16;    // storage types
    int out_bf16 = 0;         // the volume (out / grad_out) is bf16; features are fp32 then
    int feat_log2e = 0;       // forward, quad-planar copy: the staged features are multiplied by log2(e) (brick_fwd_prescales)
    // Geometry gate (AUTO on planar input, shapes both variants serve): `gate_count` points at a device counter of
    // bricks whose windows overflow LDS (k_brick_gate); a gated kernel runs when (count <= gate_limit) == wants_brick
    // and returns at once otherwise.  Null = no gate.
    const int *gate_count = nullptr;
    int gate_limit = 0;
};

inline Gate make_gate(const Problem &p, bool wants_brick) { return Gate{p.gate_count, p.gate_limit, wants_brick ? 1 : 0}; }


// Raises a kernel's dynamic-LDS limit once per (device, kernel), remembering the largest size asked for: the attribute call
// is not a stream operation, so it is kept off the per-launch path and out of graph captures after the first (warm-up)
// launch.  hipFuncSetAttribute acts on the CURRENT device's function object, hence the device in the key.
hipError_t allow_dynamic_lds(const void *kernel, size_t bytes);
// the cache key (exposed for the unit test of the key)
unsigned long long dynamic_lds_cache_key(int device, const void *kernel);

// (B*V, C, HW) -> (B*V, HW, C4), zero-padding channels C..C4; and the inverse for gradients
// (fp32 channels-last accumulator -> feature dtype, planar).
hipError_t launch_to_channels_last(const void *src, void *dst, const Problem &p, hipStream_t s);
hipError_t launch_grad_to_planar(const float *srcT, void *dst, const Problem &p, hipStream_t s);
// column-major quad-planar fp32 copy (MVHMR_LAYOUT_QUAD) -> channels-last in the feature dtype (gated like the gather kernels)
bool quad_to_channels_last_supported(const Problem &p);
hipError_t launch_quad_to_channels_last(const void *srcK, void *dst, const Problem &p, hipStream_t s);
// channels-last fp32 accumulator -> channels-last feature dtype (C4 == C required)
hipError_t launch_grad_cast(const float *srcT, void *dst, const Problem &p, hipStream_t s);

// gather variant: featT is channels-last (B,V,HW,C4) in the feature dtype
hipError_t launch_fwd_gather(const void *featT, const float *proj, const Coords &coords, void *out,
                             const Problem &p, hipStream_t s);
hipError_t launch_bwd_gather(const void *grad_out, const void *featT, const float *proj, const Coords &coords,
                             float *gradT, const Problem &p, hipStream_t s);

// brick variant (LDS-staged windows); launches return hipErrorNotSupported when the shape does not qualify.
//   forward : column-major quad-planar staged copy (launch_to_quad_planar_t), 4*nvox x (NT/128) x 32 bricks
//   backward: the same column-major quad-planar staged copy, 8 x 8 x 16 (8 x 4 x 16 for 8 views) or 4 x (NT/128) x 32 bricks
bool brick_fwd_supported(const Problem &p);
bool brick_fwd_ws_shape(const Problem &p);      // the wave-specialised forward serves this problem
bool brick_fwd_prescales(const Problem &p);     // ... and wants its staged copy multiplied by log2(e): set Problem::feat_log2e for BOTH the layout pass and the kernel
bool brick_bwd_supported(const Problem &p);
size_t brick_workspace_bytes(const Problem &p);
hipError_t launch_to_quad_planar_t(const void *src, void *dst, const Problem &p, hipStream_t s, bool brick_side = true);
// the same copy from channels-last features (BV, H, W, C)
hipError_t launch_channels_last_to_quad_planar_t(const void *src, void *dst, const Problem &p, hipStream_t s);
hipError_t launch_fwd_brick(const void *featK, const float *proj, const Coords &coords, void *out, const Problem &p,
                            hipStream_t s);

// plane backward (unproject_plane_bwd.hip): the gradient plane of one (sample, view, channel quad) accumulated in LDS, no global
// atomics; the gather family's backward for planar / quad-planar features whose maps fit (Hf * (Wf | 1) * 16 B of LDS)
bool plane_bwd_supported(const Problem &p);
size_t plane_table_bytes(const Problem &p);
hipError_t launch_bwd_plane(const void *featK, const void *grad_out, const float *proj, const Coords &coords, void *grad_features,
                            void *table, const Problem &p, hipStream_t s);

// Geometry gate.  Counts the bricks whose pooled tap windows would not fit (from the projections of each brick's 8 corner
// voxels; speed heuristic only) into *count (zeroed by the caller).  GateGeom describes the bricks and windows of the kernel
// the gate decides for.
struct GateGeom {
    int bx, by, bz;      // brick extent in voxels
    int view_group;      // views whose windows are resident together (0: all of them)
    int column_major;    // window lines run along y (forward) or x (backward)
    int parity_rows;     // forward: the rows of a window column are split by parity (line stride 2 * ceil(rows / 2), origin row even)
    int cap_slots;       // 16-B LDS slots one window set may use
    int max_chunks;      // 64-slot DMA chunks a block can issue per quad
};
GateGeom brick_fwd_gate_geom(const Problem &p);
GateGeom brick_bwd_gate_geom(const Problem &p);
hipError_t launch_brick_gate(const float *proj, const Coords &coords, int *count, const GateGeom &g, const Problem &p, hipStream_t s);
int brick_count(const Problem &p, const GateGeom &g);

// brick backward: featK quad-planar features, gradK zeroed fp32 quad-planar accumulator (same shape)
hipError_t launch_bwd_brick(const void *featK, const void *grad_out, const float *proj, const Coords &coords, float *gradK,
                            const Problem &p, hipStream_t s);
hipError_t launch_quad_grad_to_planar(const float *gradK, void *dst, const Problem &p, hipStream_t s);

// process_feature fused with the layout pass: y = W x + b as an fp32 MFMA GEMM writing the column-major quad-planar copy
bool conv1x1_quad_supported(int Cin, int Cout, int H, int W);
bool conv1x1_planar_supported(int Cin, int Cout, int HW);
bool conv1x1_wgrad_supported(int Cin, int Cout, int HW);
hipError_t launch_conv1x1_wgrad(const float *gy, const float *x, float *dW, float *db, int BV, int Cin, int Cout, int HW, hipStream_t s);
hipError_t launch_conv1x1_planar(const float *x, const float *w, const float *bias, float *dst, int BV, int Cin, int Cout, int HW, hipStream_t s);
hipError_t launch_conv1x1_quad(const float *x, const float *w, const float *bias, void *dst, int BV, int Cin, int Cout, int H, int W,
                               hipStream_t s);

// DLT triangulation of one point per sample (fp32 in / out, float64 inside); points (V,2) shared or (B,V,2) per sample
hipError_t launch_triangulate_dlt(const float *proj, const float *points, const float *conf, float *out, int B, int V, int points_per_sample,
                                  int conf_per_sample, hipStream_t s);

hipError_t launch_build_coords(float *coords_out, const float *rot, const float *center, int B, int S,
                               const double pos[3], const double sides[3], hipStream_t s);

}  // namespace mvhmr
