/*
 * mvhmr_unproject.h -- C ABI of the MI355X-native volumetric un-projection ("the hot path").
 *
 * The reference (yohanshin/MultiviewHMR) has no FFI layer: the boundary this library replaces is the
 * plain Python function
 *
 *     unprojection(features, proj_matricies, coord_volumes, aggregation_method='softmax')
 *         models/aggregation.py:20-87, sole call site models/aggregation.py:193
 *
 * and the autograd graph PyTorch builds through it (grid_sampler_2d_backward etc.).  Every entry
 * point below takes plain device pointers, sizes and a HIP stream -- no torch types -- so that any
 * host (ctypes, a torch C++ extension, a C++ trainer) can bind it; INTEGRATION.md shows the
 * reference-side stub.  multiviewhmr_amd/_capi.py is the ctypes binding the Python host side uses.
 *
 * Conventions
 *   - all pointers are DEVICE pointers on the GPU that `stream` belongs to, except the descriptor;
 *   - everything is stream-ordered: no allocation, no host synchronisation, safe to graph-capture;
 *   - scratch memory comes from the caller (mvhmr_unproject_workspace_bytes), 256-byte aligned;
 *   - tensors are dense row-major in the shapes given per function; voxel index
 *     n = (x*vol_y + y)*vol_z + z, i.e. coord_volumes[b].reshape(-1, 3)  (aggregation.py:30);
 *   - return value: MVHMR_OK or an mvhmr_status_t; mvhmr_last_error() gives the reason (thread-local).
 */
#ifndef MVHMR_UNPROJECT_H
#define MVHMR_UNPROJECT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVHMR_ABI_VERSION 4   /* 2: MVHMR_LAYOUT_QUAD became column-major (B,V,C/4,Wf,Hf,4).  3: QUAD + AUTO is geometry-gated and needs its
                                 workspace; explicit GATHER with QUAD input is served; MVHMR_BF16 out_dtype; mvhmr_unproject_backward_supported,
                                 mvhmr_triangulate_dlt.  4: MVHMR_LAYOUT_QUAD_LOG2E (ABI history) */

typedef enum mvhmr_status_t {
    MVHMR_OK = 0,
    MVHMR_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive size, unknown enum (-> ValueError/RuntimeError in Python) */
    MVHMR_ERR_UNSUPPORTED = 2,      /* legal request this build has no kernel for */
    MVHMR_ERR_WORKSPACE = 3,        /* workspace missing, too small or misaligned */
    MVHMR_ERR_LAUNCH = 4            /* hipGetLastError() after a launch was not hipSuccess */
} mvhmr_status_t;

/* aggregation_method of models/aggregation.py:71-85 */
typedef enum mvhmr_agg_t {
    MVHMR_AGG_SOFTMAX = 0, /* sum_v x_v * softmax_v(x)_v  (aggregation.py:77-83) */
    MVHMR_AGG_SUM = 1,     /* aggregation.py:71-72 */
    MVHMR_AGG_MEAN = 2,    /* aggregation.py:73-74 : divides by V, masked views included */
    MVHMR_AGG_MAX = 3      /* aggregation.py:75-76 */
} mvhmr_agg_t;

/* storage type of features / out / grad_out / grad_features.  Coordinates, projection matrices, tap
 * weights, the cross-view softmax and all accumulation are always fp32.  The reference is fp32 only;
 * MVHMR_F16 is this library's storage mode (SURVEY.md 8d "fp16 convention"). */
typedef enum mvhmr_dtype_t {
    MVHMR_F32 = 0,
    MVHMR_F16 = 1,
    MVHMR_BF16 = 2  /* out_dtype only, with fp32 features: the volume (and grad_out) stored as bf16 for a half-precision consumer
                       (models/regressor.py:70-87 under autocast); round to nearest even at the store, NaN stays NaN */
} mvhmr_dtype_t;

/* memory layout of `features` (and of `grad_features`) */
typedef enum mvhmr_layout_t {
    MVHMR_LAYOUT_BVCHW = 0, /* (B,V,C,Hf,Wf) -- the reference's contract (aggregation.py:22-23, :191) */
    MVHMR_LAYOUT_BVHWC = 1, /* (B,V,Hf,Wf,C) -- channels-last, what the gather variant reads; passing it
                               skips the layout pass (e.g. a channels_last 1x1 conv upstream) */
    MVHMR_LAYOUT_QUAD = 2,  /* (B,V,C/4,Wf,Hf,4) fp32 whatever feat_dtype -- column-major "quad-planar": a pixel's 4 channels
                               are 16 contiguous bytes and a pixel COLUMN is one contiguous run, which is what the brick
                               forward stages into LDS (tall narrow tap windows); produced by mvhmr_convert_features and by
                               mvhmr_conv1x1_to_quad.  C % 4 != 0 (round 5): (C + 3) / 4 quads per view, the last one zero-padded
                               by mvhmr_convert_features; brick kernels only (their loops run the whole quads, the last 1 ... 3
                               channels go per voxel).  Forward and backward accept it; the backward then writes
                               grad_features PLANAR (B,V,C,Hf,Wf).  With MVHMR_VARIANT_AUTO the geometry gate decides on the
                               device as for planar input (the gather side converts the copy to channels-last first: C % 4 == 0) */
    MVHMR_LAYOUT_QUAD_LOG2E = 3 /* the same copy with every value multiplied by log2(e), FORWARD ONLY: what the wave-specialised softmax
                               forward stages (3 / 4 views, fp32 volume, launches of >= 256 bricks): its exponentials are then exp2 of
                               a plain difference and ln 2 is folded into the final multiply (<= 2e-7 relative to the unscaled
                               route).  mvhmr_preferred_layout returns it exactly when the forward accepts it; the backward and every
                               other shape / aggregate answer MVHMR_ERR_UNSUPPORTED.  A forward on planar input makes this copy itself */
} mvhmr_layout_t;

/* kernel selection; AUTO picks the fastest applicable one.  The others exist for tests and profiling.
 * AUTO with planar or quad-planar features of a shape both variants serve decides ON THE DEVICE (cameras and voxel pitch decide whether
 * the brick variant's LDS windows fit): both variants are launched behind a gate and one of them runs; stream-ordered, no
 * host synchronisation.  mvhmr_unproject_selected_variant reports the variant AUTO prefers for the shape. */
typedef enum mvhmr_variant_t {
    MVHMR_VARIANT_AUTO = 0,
    MVHMR_VARIANT_GATHER = 1, /* channel-per-lane gather from L2, any shape */
    MVHMR_VARIANT_BRICK = 2   /* voxel bricks with LDS-staged feature windows (forward) and LDS-accumulated
                                 gradient windows (backward): C % 4 == 0, 1 ... 8 views (1, 3 and 5 / 6 / 7 views run the 2-, 4- and 8-view
                                 kernels with the missing views absent), any volume extents (forward bricks of
                                 4 x 8 x 32 voxels, 8 x 8 x 32 for 2 / 4 views when vol_x > 4; backward 8 x 8 x 16 -- 8 x 4 x 16 with
                                 8 views -- or 4 x 8 x 32 / 4 x 4 x 32, whichever covers the volume with fewer idle lanes; the lanes
                                 of a brick that lie past the volume's edge idle; 16-bit volumes need vol_z even in the forward);
                                 every storage pairing check_desc admits (fp32 / fp16 features with an fp32, fp16 or -- fp32 features -- bf16 volume);
                                 anything else is MVHMR_ERR_UNSUPPORTED.
                                 What therefore runs the GATHER family under AUTO: more than 8 views; C % 4 != 0;
                                 channels-last input; forward launches of fewer than 96 bricks' worth of voxels (B * X * Y * Z <
                                 196 608: one round of bricks costs the same however few they are -- the reference's shipped
                                 16^3 volume, cfg/defaults.py:25, up to batch 47 per GPU); and any call whose cameras / voxel
                                 pitch make the LDS windows overflow (decided on the device).  Its backward is the plane kernel (no
                                 global atomics, 64-bit sums where many taps meet in a pixel) for planar / quad-planar features whose
                                 maps fit LDS, the per-tap float scatter otherwise; AUTO also prefers the plane kernel to bricks when
                                 the volume has fewer bricks than the chip has CUs */
} mvhmr_variant_t;

typedef struct mvhmr_unproject_desc {
    int32_t abi_version; /* MVHMR_ABI_VERSION */
    int32_t batch;       /* B  = features.shape[0]              */
    int32_t views;       /* V  = features.shape[1]   (1..16)    */
    int32_t channels;    /* C  = features.shape[2]              */
    int32_t feat_h;      /* Hf = features.shape[3]              */
    int32_t feat_w;      /* Wf = features.shape[4]              */
    int32_t vol_x;       /* coord_volumes.shape[1]              */
    int32_t vol_y;       /* coord_volumes.shape[2]              */
    int32_t vol_z;       /* coord_volumes.shape[3]              */
    int32_t method;      /* mvhmr_agg_t                         */
    int32_t feat_dtype;  /* mvhmr_dtype_t of features / grad_features */
    int32_t out_dtype;   /* mvhmr_dtype_t of out / grad_out (the reference always returns fp32, aggregation.py:25) */
    int32_t feat_layout; /* mvhmr_layout_t                      */
    int32_t variant;     /* mvhmr_variant_t                     */
} mvhmr_unproject_desc;

/* Name of the kernel family the forward launches for this descriptor ("k_fwd_ws", "k_fwd_brick", "k_fwd_brick_groups", "k_fwd_gather";
 * for MVHMR_VARIANT_AUTO on gated shapes: the brick-side kernel the device-side gate may select): lets a profiler harness match
 * rocprofv3 kernel names without knowing the dispatch rules.  Static storage; NULL for an invalid descriptor. */
const char *mvhmr_unproject_forward_kernel_name(const mvhmr_unproject_desc *desc);

/* 1 when mvhmr_unproject_backward[_cuboid] serves this descriptor (layout x variant x shape), else 0: lets set-up code choose a
 * route before any tensor exists (the forward's counterpart is mvhmr_unproject_selected_variant > 0). */
int mvhmr_unproject_backward_supported(const mvhmr_unproject_desc *desc);

/* Bytes of device scratch forward / backward need for this problem (0 is possible).  Forward: at most one fp32 copy of the features.
 * Backward: a feature copy + an fp32 gradient accumulator of the same size on the brick route; on the coarse-grid (plane) route a tap
 * table and one slab of the Jacobian stream (V x the slab's share of grad_out in fp32, at most ~256 MB: 0.27 GB for BASELINE configs[1]) --
 * ask, do not guess. */
size_t mvhmr_unproject_forward_workspace_bytes(const mvhmr_unproject_desc *desc);
size_t mvhmr_unproject_backward_workspace_bytes(const mvhmr_unproject_desc *desc);

/*
 * Forward: replaces unprojection() (models/aggregation.py:20-87).
 *   features  (B,V,C,Hf,Wf) or (B,V,Hf,Wf,C) per desc->feat_layout, desc->feat_dtype   [read]
 *   proj      (B,V,3,4) fp32   -- proj_matricies (aggregation.py:132-133)              [read]
 *   coords    (B,X,Y,Z,3) fp32 -- coord_volumes  (aggregation.py:136,187)              [read]
 *   out       (B,C,X,Y,Z) desc->out_dtype, every element is written                     [write]
 */
int mvhmr_unproject_forward(const mvhmr_unproject_desc *desc, const void *features, const float *proj,
                            const float *coords, void *out, void *workspace, size_t workspace_bytes,
                            void *hip_stream);

/*
 * Backward w.r.t. features: replaces autograd through the reference graph (CopySlices, softmax/mul/sum,
 * masked fill, grid_sampler_2d_backward per (b, v)); proj and coords never need gradients
 * (they are built from numpy / arange, aggregation.py:132-187).
 *   grad_out       (B,C,X,Y,Z) desc->out_dtype                                          [read]
 *   grad_features  same shape/layout/dtype as features, every element is written        [write]
 * Scatter-adds use fp32 float atomics, so low-order bits can differ from run to run.  The brick variant sums a
 * brick's contributions per pixel in fixed point first (one power-of-two scale per channel, resolution ~2^-25 of the
 * channel's largest contribution in the brick) and issues one float atomic per window pixel.
 * Non-finite values: an Inf / NaN in grad_out (or an overflowing product) makes grad_features non-finite.  The gather variant
 * propagates it tap by tap like the reference's float scatter; the brick variant writes NaN to every pixel of the affected
 * brick's tap windows for that channel (a superset of the reference's pixels) -- isfinite(grad) agrees, the exact set differs.
 * Forward: a tap outside the image contributes 0 * (clamped border pixel) instead of being skipped.  With a NON-FINITE feature
 * value in a border pixel the same samples come out non-finite as with grid_sample's zero padding (such a sample also taps the
 * border pixel itself), but an Inf may read NaN; finite outputs are never affected.  A NaN depth (NaN in proj / coords) gives
 * an exactly zero sample for that view.
 */
int mvhmr_unproject_backward(const mvhmr_unproject_desc *desc, const void *grad_out, const void *features,
                             const float *proj, const float *coords, void *grad_features, void *workspace,
                             size_t workspace_bytes, void *hip_stream);

/*
 * The same two calls for the volumes VolumeGenerator.forward builds (models/aggregation.py:138-187): instead of reading a
 * (B,X,Y,Z,3) coordinate tensor the kernels evaluate the reference's recipe per voxel,
 *     coords[b,i,j,k] = rot[b] @ (position + sides / (S - 1) * (i,j,k) - center[b]) + center[b],      S = (vol_x, vol_y, vol_z)
 * with the rounding order of mvhmr_build_coord_volumes, so that both routes are bit-equal.  13 floats per sample replace the
 * coordinate tensor (403 MB per GPU at BASELINE configs[4]) and the kernel that builds it.
 *   rot       (B,9) fp32 row-major, device   -- utils/volumetric.py:87-114 (identity at eval)
 *   center    (B,3) fp32, device             -- the rotation pivot, aggregation.py:180-186
 *   position, sides   3 doubles each, HOST   -- cuboid corner and edge lengths, aggregation.py:143-144
 */
int mvhmr_unproject_forward_cuboid(const mvhmr_unproject_desc *desc, const void *features, const float *proj,
                                   const float *rot, const float *center, const double position[3],
                                   const double sides[3], void *out, void *workspace, size_t workspace_bytes,
                                   void *hip_stream);
int mvhmr_unproject_backward_cuboid(const mvhmr_unproject_desc *desc, const void *grad_out, const void *features,
                                    const float *proj, const float *rot, const float *center,
                                    const double position[3], const double sides[3], void *grad_features,
                                    void *workspace, size_t workspace_bytes, void *hip_stream);

/*
 * Layout pass on its own: features (B,V,C,Hf,Wf) -> dst in `dst_layout` (MVHMR_LAYOUT_BVHWC with the channel
 * count rounded up to a multiple of 4 and zero padded, or MVHMR_LAYOUT_QUAD), desc->feat_dtype.
 * desc->feat_layout names the SOURCE: MVHMR_LAYOUT_BVCHW (also assumed for MVHMR_LAYOUT_QUAD descriptors, as before), or
 * MVHMR_LAYOUT_BVHWC -- channels-last features to MVHMR_LAYOUT_QUAD only (how a channels-last caller reaches the brick kernels).
 * mvhmr_unproject_forward runs the pass its kernel needs into its workspace when desc->feat_layout is
 * MVHMR_LAYOUT_BVCHW; callers that keep the converted copy (or time the two kernels separately) call this and
 * then pass desc->feat_layout = dst_layout.  mvhmr_preferred_layout says which layout the kernel that
 * desc->variant selects reads; dst needs mvhmr_feature_layout_bytes(desc, dst_layout) bytes.
 */
int mvhmr_preferred_layout(const mvhmr_unproject_desc *desc);
size_t mvhmr_feature_layout_bytes(const mvhmr_unproject_desc *desc, int dst_layout);
int mvhmr_convert_features(const mvhmr_unproject_desc *desc, const void *features, int dst_layout, void *dst,
                           void *hip_stream);

/*
 * process_feature (the 1x1 conv in front of the un-projection, models/aggregation.py:108-110,189-191) fused with the layout pass:
 *     y[m, co, p] = sum_ci weight[co, ci] * x[m, ci, p] + bias[co]          m = (b, v),  p = (y, x)
 * computed as an fp32 MFMA GEMM whose epilogue writes dst in MVHMR_LAYOUT_QUAD (n_maps, c_out/4, Wf, Hf, 4) -- feed it to
 * mvhmr_unproject_forward[_cuboid] with desc->feat_layout = MVHMR_LAYOUT_QUAD and the planar conv output and the layout pass
 * never exist (variant AUTO: the brick kernels read the copy as it is; when the geometry gate picks the gather kernels they get a
 * channels-last conversion of it).  mvhmr_unproject_backward[_cuboid] accepts the same MVHMR_LAYOUT_QUAD features and
 * then writes grad_features in the PLANAR layout (n_maps, c_out, Hf, Wf), which is what the conv's own backward consumes.
 *   x (n_maps, c_in, Hf, Wf) fp32, weight (c_out, c_in) fp32 (nn.Conv2d's (c_out, c_in, 1, 1)), bias (c_out) fp32 or null.
 * Shapes: c_in % 16 == 0, c_out % 128 == 0, Hf % 4 == 0, Wf % 32 == 0 (mvhmr_conv1x1_to_quad_supported), else MVHMR_ERR_UNSUPPORTED.
 */
int mvhmr_conv1x1_to_quad(const float *x, const float *weight, const float *bias, void *dst, int32_t n_maps, int32_t c_in,
                          int32_t c_out, int32_t feat_h, int32_t feat_w, void *hip_stream);
int mvhmr_conv1x1_to_quad_supported(int32_t c_in, int32_t c_out, int32_t feat_h, int32_t feat_w);

/*
 * The same fp32 MFMA GEMM with a planar result: dst (n_maps, c_out, pixels) = weight (c_out, c_in) @ x (n_maps, c_in, pixels)
 * (+ bias).  With the transposed weight it is the gradient w.r.t. the input of process_feature (autograd through
 * models/aggregation.py:189-191), which the Python binding's fused route uses in backward.
 * Shapes: c_in % 16 == 0, c_out % 128 == 0, pixels % 128 == 0 (mvhmr_conv1x1_planar_supported), else MVHMR_ERR_UNSUPPORTED.
 */
int mvhmr_conv1x1_planar(const float *x, const float *weight, const float *bias, float *dst, int32_t n_maps, int32_t c_in,
                         int32_t c_out, int32_t pixels, void *hip_stream);
int mvhmr_conv1x1_planar_supported(int32_t c_in, int32_t c_out, int32_t pixels);

/*
 * Weight and bias gradient of process_feature (autograd through models/aggregation.py:189-191):
 *     grad_weight[co, ci] += sum over maps and pixels of grad_y[n, co, p] * x[n, ci, p]      grad_bias[co] += sum of grad_y[n, co, p]
 * grad_y (n_maps, c_out, pixels), x (n_maps, c_in, pixels) fp32 planar.  grad_weight (c_out, c_in) and grad_bias (c_out, may be
 * NULL) are ADDED INTO with float atomics (zero them first; last-bit run-to-run differences like any split-K reduction).
 * Shapes: c_in % 128 == 0, c_out % 128 == 0, pixels % 32 == 0 (mvhmr_conv1x1_wgrad_supported), else MVHMR_ERR_UNSUPPORTED.
 */
int mvhmr_conv1x1_wgrad(const float *grad_y, const float *x, float *grad_weight, float *grad_bias, int32_t n_maps, int32_t c_in,
                        int32_t c_out, int32_t pixels, void *hip_stream);
int mvhmr_conv1x1_wgrad_supported(int32_t c_in, int32_t c_out, int32_t pixels);

/*
 * DLT triangulation of one 3-D point per sample from its V views: replaces triangulate_point_from_multiple_views_linear[_torch]
 * (utils/multiview.py:112-168) where VolumeGenerator.forward calls it per sample with a device SVD and a .cpu() each
 * (models/aggregation.py:174-177).  Homogeneous solution of A h = 0 (rows u P[2,:] - P[0,:], v P[2,:] - P[1,:]) as the smallest
 * eigenvector of the normal matrix A^T A (the unit-norm least-squares solution, as the SVD gives it), float64 Jacobi rotations, one
 * thread per sample, no host synchronisation.
 *   proj    (B,V,3,4) fp32, device      points  (V,2) fp32 shared by the samples (points_per_sample = 0) or (B,V,2) (= 1), device
 *   out     (B,3) fp32, device
 */
int mvhmr_triangulate_dlt(const float *proj, const float *points, float *out, int32_t batch, int32_t views,
                          int32_t points_per_sample, void *hip_stream);
/* the same with the per-view confidences of triangulate_point_from_multiple_views_linear_torch (utils/multiview.py:156-161: both rows of
 * view v are multiplied by c_v before the decomposition): confidences (V) fp32 shared by the samples (confidences_per_sample = 0) or (B,V) */
int mvhmr_triangulate_dlt_weighted(const float *proj, const float *points, const float *confidences, float *out, int32_t batch, int32_t views,
                                   int32_t points_per_sample, int32_t confidences_per_sample, void *hip_stream);

/*
 * Caller-side helper of VolumeGenerator.forward (models/aggregation.py:138-187): fills
 * coords (B,S,S,S,3) fp32 with the cuboid grid `position + side/(S-1) * (i,j,k)` rotated by
 * rot[b] (3x3 row-major fp32, utils/volumetric.py:87-114) about center[b] (3 fp32):
 *     coords = rot @ (grid - center) + center
 * position is the cuboid corner, sides its edge lengths (3 doubles each, HOST memory, as the reference
 * holds them in float64 numpy, aggregation.py:143-144); rot / center / coords are device pointers.
 */
int mvhmr_build_coord_volumes(float *coords, const float *rot, const float *center, int32_t batch,
                              int32_t volume_size, const double position[3], const double sides[3],
                              void *hip_stream);

/* Which kernel AUTO would run for this problem (an mvhmr_variant_t), for logs and tests. */
int mvhmr_unproject_selected_variant(const mvhmr_unproject_desc *desc);

/*
 * Planning query for callers that run the layout pass themselves (mvhmr_convert_features + an explicit layout, which the
 * device-side gate does not cover): the variant (mvhmr_variant_t) the gate would select for THIS geometry, or -1 on error.
 * SYNCHRONOUS -- allocates 4 bytes, runs the gate kernel on hip_stream and waits for it: for set-up code, never for the
 * per-step path.  proj (B,V,3,4) and coords (B,X,Y,Z,3) as for mvhmr_unproject_forward.
 */
int mvhmr_unproject_query_variant(const mvhmr_unproject_desc *desc, const float *proj, const float *coords,
                                  void *hip_stream);
int mvhmr_unproject_query_variant_cuboid(const mvhmr_unproject_desc *desc, const float *proj, const float *rot,
                                         const float *center, const double position[3], const double sides[3],
                                         void *hip_stream);

/* Testing hook: the key under which the library remembers that it raised a kernel's dynamic-LDS limit (the attribute is
 * per device AND kernel; a second GPU driven from the same process must get its own opt-in). */
unsigned long long mvhmr_internal_lds_cache_key(int device, const void *kernel);

int mvhmr_abi_version(void);
const char *mvhmr_status_string(int status);
const char *mvhmr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MVHMR_UNPROJECT_H */
