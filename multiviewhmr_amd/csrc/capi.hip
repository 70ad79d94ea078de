// extern "C" surface of libmvhmr_unproject.so -- see include/mvhmr_unproject.h for the contract and the
// reference interface (models/aggregation.py:20-87) each entry point replaces.
#include "mvhmr_unproject.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <unordered_map>

#include "device_common.h"
#include "kernels.h"

using namespace mvhmr;

namespace mvhmr {
unsigned long long dynamic_lds_cache_key(int device, const void *kernel)
{
    // kernel entry points are at least 256-B aligned, so the low byte of the pointer is free for the device ordinal
    return (unsigned long long)reinterpret_cast<uintptr_t>(kernel) ^ ((unsigned long long)(unsigned)device << 56) ^ (unsigned long long)(device & 0xff);
}

hipError_t allow_dynamic_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::unordered_map<unsigned long long, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long key = dynamic_lds_cache_key(dev, kernel);
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[key] = bytes;
    return e;
}
}  // namespace mvhmr

namespace {

thread_local char g_err[512] = "";

int fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}

constexpr size_t kAlign = 256;
size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

int check_desc(const mvhmr_unproject_desc *d, Problem *p)
{
    if (!d) return fail(MVHMR_ERR_INVALID_ARGUMENT, "descriptor is null");
    if (d->abi_version != MVHMR_ABI_VERSION)
        return fail(MVHMR_ERR_INVALID_ARGUMENT, "descriptor abi_version %d, library speaks %d", d->abi_version, MVHMR_ABI_VERSION);
    if (d->batch < 1 || d->views < 1 || d->channels < 1 || d->feat_h < 1 || d->feat_w < 1 || d->vol_x < 1 || d->vol_y < 1 || d->vol_z < 1)
        return fail(MVHMR_ERR_INVALID_ARGUMENT, "every dimension must be >= 1 (B=%d V=%d C=%d Hf=%d Wf=%d vol=%dx%dx%d)", d->batch, d->views,
                    d->channels, d->feat_h, d->feat_w, d->vol_x, d->vol_y, d->vol_z);
    if (d->method < MVHMR_AGG_SOFTMAX || d->method > MVHMR_AGG_MAX)
        return fail(MVHMR_ERR_INVALID_ARGUMENT, "Unknown aggregation_method: %d", d->method);
    if (d->feat_dtype < 0 || d->feat_dtype > MVHMR_BF16 || d->out_dtype < 0 || d->out_dtype > MVHMR_BF16)
        return fail(MVHMR_ERR_INVALID_ARGUMENT, "unknown dtype (feat %d, out %d)", d->feat_dtype, d->out_dtype);
    if (d->feat_dtype == MVHMR_BF16)
        return fail(MVHMR_ERR_UNSUPPORTED, "bf16 is a storage type of the volume only (out_dtype); features are fp32 or fp16");
    if (d->out_dtype == MVHMR_BF16 && d->feat_dtype != MVHMR_F32)
        return fail(MVHMR_ERR_UNSUPPORTED, "a bf16 volume needs fp32 features");
    if (d->feat_layout < 0 || d->feat_layout > MVHMR_LAYOUT_QUAD_LOG2E) return fail(MVHMR_ERR_INVALID_ARGUMENT, "unknown feature layout %d", d->feat_layout);
    if (d->variant < 0 || d->variant > MVHMR_VARIANT_BRICK) return fail(MVHMR_ERR_INVALID_ARGUMENT, "unknown kernel variant %d", d->variant);
    if (d->views > kMaxViews) return fail(MVHMR_ERR_UNSUPPORTED, "at most %d views are supported (got %d)", kMaxViews, d->views);
    if (d->feat_dtype == MVHMR_F32 && d->out_dtype == MVHMR_F16)
        return fail(MVHMR_ERR_UNSUPPORTED, "fp32 features with an fp16 volume is not a supported storage mode");
    p->B = d->batch; p->V = d->views; p->C = d->channels; p->H = d->feat_h; p->W = d->feat_w;
    p->X = d->vol_x; p->Y = d->vol_y; p->Z = d->vol_z;
    p->N = (long long)d->vol_x * d->vol_y * d->vol_z;
    p->C4 = (d->channels + 3) / 4 * 4;
    p->method = d->method;
    p->feat_f16 = d->feat_dtype == MVHMR_F16;
    p->out_f16 = d->out_dtype == MVHMR_F16;
    p->out_bf16 = d->out_dtype == MVHMR_BF16;
    if ((long long)p->H * p->W * p->C4 >= (1ll << 31))
        return fail(MVHMR_ERR_UNSUPPORTED, "one feature map (Hf*Wf*C = %lld elements) exceeds 32-bit tap offsets", (long long)p->H * p->W * p->C4);
    if (d->feat_layout == MVHMR_LAYOUT_BVHWC && p->C4 != p->C)
        return fail(MVHMR_ERR_UNSUPPORTED, "channels-last features need C %% 4 == 0 (C = %d)", p->C);
    // (quad-planar copies hold (C + 3) / 4 quads per view, the last one zero-padded: any C the brick kernels take)
    return MVHMR_OK;
}

size_t feat_elem(const Problem &p) { return p.feat_f16 ? 2 : 4; }
size_t featT_bytes(const Problem &p) { return align_up((size_t)p.B * p.V * p.H * p.W * p.C4 * feat_elem(p)); }
size_t gradT_bytes(const Problem &p) { return align_up((size_t)p.B * p.V * p.H * p.W * p.C4 * sizeof(float)); }

constexpr int kChipCUs = 256;       // MI355X

// AUTO prefers the brick forward only when the launch holds enough voxels to keep the chip busy: every block walks all C / 4 quads
// whatever the volume, so one round of bricks costs the same run time however few of them there are, while the gather kernels' time
// falls with the voxel count.  Measured break-even (profiles/r04_configs/brick_count_sweep.txt: 16^3 ... 32^3 grids, 32 ... 256
// channels, 12 ... 96 px maps): between 64 and 108 bricks' worth of voxels; the reference's shipped 16^3 x 32 samples (64 bricks' worth,
// half of every brick past the volume's top) runs 0.13 ms on the gather kernels against 0.21 ms.  variant = brick still forces the bricks.
constexpr long long kBrickFwdMinVoxels = 96ll * 8 * 8 * 32;             // 96 bricks of 8 x 8 x 32
bool brick_fwd_preferred(const Problem &p)
{
    return brick_fwd_supported(p) && static_cast<long long>(p.B) * p.X * p.Y * p.Z >= kBrickFwdMinVoxels;
}

// which kernel runs: channels-last input feeds the gather kernels; quad-planar input (the fused 1x1 conv's output) the brick kernels,
// or -- through one more layout pass -- the gather kernels when the caller or the shape asks for them
int pick_variant(const mvhmr_unproject_desc *d, const Problem &p)
{
    if (d->feat_layout == MVHMR_LAYOUT_BVHWC) return MVHMR_VARIANT_GATHER;
    if (d->variant != MVHMR_VARIANT_AUTO) return d->variant;
    return brick_fwd_preferred(p) ? MVHMR_VARIANT_BRICK : MVHMR_VARIANT_GATHER;
}

int variant_conflict(const mvhmr_unproject_desc *d, const Problem &p, int variant)
{
    if (variant == MVHMR_VARIANT_BRICK && !brick_fwd_supported(p)) return fail(MVHMR_ERR_UNSUPPORTED, "the brick variant does not support this shape / dtype");
    if (d->variant != MVHMR_VARIANT_AUTO && d->variant != variant)
        return fail(MVHMR_ERR_UNSUPPORTED, "feature layout %d cannot feed kernel variant %d", d->feat_layout, d->variant);
    if (variant == MVHMR_VARIANT_GATHER && d->feat_layout == MVHMR_LAYOUT_QUAD && !quad_to_channels_last_supported(p))
        return fail(MVHMR_ERR_UNSUPPORTED, "quad-planar features of this shape cannot feed the gather kernels (C <= 4092, B * V <= 65535)");
    return MVHMR_OK;
}

// the gradient can be accumulated straight into grad_features when that already is fp32 channels-last
bool grad_in_place(const mvhmr_unproject_desc *d, const Problem &p) { return d->feat_layout == MVHMR_LAYOUT_BVHWC && !p.feat_f16; }

// The brick backward (window gradients accumulated in LDS in fixed point, then flushed with 256-B shaped float atomics)
// is the default wherever the brick forward is: ~20 GB of global atomic traffic instead of the gather backward's 137 GB
// (17 ms against 104 ms at the north-star size: profiles/r01_final_pmc.txt).  variant = gather keeps the gather backward.
bool bwd_uses_brick(const mvhmr_unproject_desc *d, const Problem &p)
{
    if (d->feat_layout == MVHMR_LAYOUT_QUAD && p.feat_f16) return false;        // the quad copy is fp32; mixed storage goes through the gather backward
    // A volume of fewer bricks than the chip has CUs (the reference's shipped 16^3: 4 bricks per sample) leaves most of it idle while every
    // block still walks all C / 4 quads: 0.48 ms for 32 samples of 16^3 against 0.2x for the plane kernels, whose blocks are (sample,
    // view, quad).  AUTO then takes the plane backward; variant = brick still forces the bricks.
    if (d->variant == MVHMR_VARIANT_AUTO && plane_bwd_supported(p) && brick_bwd_supported(p) && brick_count(p, brick_bwd_gate_geom(p)) < kChipCUs) return false;
    return (d->feat_layout == MVHMR_LAYOUT_BVCHW || d->feat_layout == MVHMR_LAYOUT_QUAD) && d->variant != MVHMR_VARIANT_GATHER && brick_bwd_supported(p);
}

// The gather family's backward for planar / quad-planar features: the plane kernel (no global atomics) where the maps fit LDS, else
// the per-tap scatter.  Channels-last features keep the scatter (its accumulator is the caller's own tensor).
bool bwd_uses_plane(const mvhmr_unproject_desc *d, const Problem &p)
{
    return d->feat_layout != MVHMR_LAYOUT_BVHWC && plane_bwd_supported(p);
}
// bytes between the converted feature copy and the gate counter of a gated backward: the brick side's accumulator or the plane
// side's tap table, whichever is larger
size_t bwd_mid_bytes(const mvhmr_unproject_desc *d, const Problem &p)
{
    const size_t a = gradT_bytes(p), b = bwd_uses_plane(d, p) ? align_up(plane_table_bytes(p)) : 0;
    return a > b ? a : b;
}

// AUTO on planar input, for a shape both variants serve: the variant is chosen on the device from the geometry (gate.h).
// Quad-planar input is gated the same way (the gather side then converts it to channels-last first), so a caller that keeps
// only the fused conv's copy never pins a variant the geometry does not suit.
bool gateable_layout(const mvhmr_unproject_desc *d, const Problem &p)
{
    return d->feat_layout == MVHMR_LAYOUT_BVCHW || (d->feat_layout == MVHMR_LAYOUT_QUAD && quad_to_channels_last_supported(p));
}
bool geometry_gated(const mvhmr_unproject_desc *d, const Problem &p)
{
    return d->variant == MVHMR_VARIANT_AUTO && gateable_layout(d, p) && brick_fwd_preferred(p);
}
// the backward has its own bricks and windows, hence its own gate
bool geometry_gated_bwd(const mvhmr_unproject_desc *d, const Problem &p)
{
    return d->variant == MVHMR_VARIANT_AUTO && gateable_layout(d, p) && brick_bwd_supported(p);
}
constexpr size_t kGateBytes = 256;
// the converted feature copy of a gated launch: channels-last in the feature dtype or quad-planar fp32, whichever is larger
size_t conv_bytes(const Problem &p) { const size_t a = featT_bytes(p), b = brick_workspace_bytes(p); return a > b ? a : b; }

// zeroes the counter at the end of the workspace region `at`, counts the overflowing bricks, arms the gate in p
int arm_gate(Problem &p, unsigned char *at, const float *proj, const Coords &coords, const GateGeom &g, hipStream_t s)
{
    int *count = reinterpret_cast<int *>(at);
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
    if (e != hipSuccess) return fail(MVHMR_ERR_LAUNCH, "geometry gate clear: %s", hipGetErrorString(e));
    e = launch_brick_gate(proj, coords, count, g, p, s);
    if (e != hipSuccess) return fail(MVHMR_ERR_LAUNCH, "geometry gate: %s", hipGetErrorString(e));
    p.gate_count = count;
    p.gate_limit = brick_count(p, g) / 8;       // brick variant while at most 1/8 of the bricks would take its slow path
    return MVHMR_OK;
}

int check_ws(void *ws, size_t have, size_t need)
{
    if (need == 0) return MVHMR_OK;
    if (!ws) return fail(MVHMR_ERR_WORKSPACE, "workspace is null but %zu bytes are required", need);
    if (have < need) return fail(MVHMR_ERR_WORKSPACE, "workspace has %zu bytes, %zu are required", have, need);
    if (reinterpret_cast<uintptr_t>(ws) % kAlign) return fail(MVHMR_ERR_WORKSPACE, "workspace must be %zu-byte aligned", kAlign);
    return MVHMR_OK;
}

Coords coords_from_tensor(const float *coords, const Problem &p)
{
    Coords c{};
    c.ptr = coords; c.Y = p.Y; c.Z = p.Z;
    return c;
}

// the reference's cuboid recipe (aggregation.py:140-187): corner `position`, edge lengths `sides` (float64 on the host, cast to
// fp32 when they meet the tensor), per-sample rotation and pivot on the device
int coords_from_cuboid(const float *rot, const float *center, const double position[3], const double sides[3], const Problem &p, Coords *out)
{
    if (!rot || !center || !position || !sides) return fail(MVHMR_ERR_INVALID_ARGUMENT, "rot / center / position / sides must be non-null");
    Coords c{};
    c.ptr = nullptr; c.rot = rot; c.center = center; c.Y = p.Y; c.Z = p.Z;
    c.px = (float)position[0]; c.py = (float)position[1]; c.pz = (float)position[2];
    // step = sides / (S - 1) in float64, then fp32 (aggregation.py:157-159); a one-voxel axis has no step
    c.sx = p.X > 1 ? (float)(sides[0] / (double)(p.X - 1)) : 0.f;
    c.sy = p.Y > 1 ? (float)(sides[1] / (double)(p.Y - 1)) : 0.f;
    c.sz = p.Z > 1 ? (float)(sides[2] / (double)(p.Z - 1)) : 0.f;
    *out = c;
    return MVHMR_OK;
}

int launched(hipError_t e, const char *what)
{
    if (e == hipSuccess) return MVHMR_OK;
    if (e == hipErrorNotSupported) return fail(MVHMR_ERR_UNSUPPORTED, "%s: no kernel for this dtype/shape combination", what);
    return fail(MVHMR_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace

extern "C" {

int mvhmr_abi_version(void) { return MVHMR_ABI_VERSION; }

unsigned long long mvhmr_internal_lds_cache_key(int device, const void *kernel) { return dynamic_lds_cache_key(device, kernel); }

const char *mvhmr_last_error(void) { return g_err; }

const char *mvhmr_status_string(int status)
{
    switch (status) {
    case MVHMR_OK: return "ok";
    case MVHMR_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MVHMR_ERR_UNSUPPORTED: return "unsupported";
    case MVHMR_ERR_WORKSPACE: return "workspace";
    case MVHMR_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
    }
}

int mvhmr_unproject_selected_variant(const mvhmr_unproject_desc *desc)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK) return -1;
    const int variant = pick_variant(desc, p);
    return variant_conflict(desc, p, variant) == MVHMR_OK ? variant : -1;
}

int mvhmr_unproject_query_variant(const mvhmr_unproject_desc *desc, const float *proj, const float *coords, void *hip_stream)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK) return -1;
    const int variant = pick_variant(desc, p);
    if (variant_conflict(desc, p, variant) != MVHMR_OK) return -1;
    if (desc->variant != MVHMR_VARIANT_AUTO || !brick_fwd_preferred(p)) return variant;   // nothing to decide
    if (!proj || !coords) { fail(MVHMR_ERR_INVALID_ARGUMENT, "proj / coords must be non-null"); return -1; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int *count = nullptr, host = 0;
    if (hipMalloc(&count, sizeof(int)) != hipSuccess) { fail(MVHMR_ERR_LAUNCH, "query: allocation failed"); return -1; }
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
    const GateGeom g = brick_fwd_gate_geom(p);
    if (e == hipSuccess) e = launch_brick_gate(proj, coords_from_tensor(coords, p), count, g, p, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&host, count, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(count);
    if (e != hipSuccess) { fail(MVHMR_ERR_LAUNCH, "query: %s", hipGetErrorString(e)); return -1; }
    return host <= brick_count(p, g) / 8 ? MVHMR_VARIANT_BRICK : MVHMR_VARIANT_GATHER;
}

const char *mvhmr_unproject_forward_kernel_name(const mvhmr_unproject_desc *desc)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK) return nullptr;
    const int variant = pick_variant(desc, p);
    if (variant_conflict(desc, p, variant) != MVHMR_OK) return nullptr;
    if (variant != MVHMR_VARIANT_BRICK) return "k_fwd_gather";
    if (brick_fwd_ws_shape(p)) return "k_fwd_ws";
    return p.V > 4 ? "k_fwd_brick_groups" : "k_fwd_brick";
}

int mvhmr_unproject_backward_supported(const mvhmr_unproject_desc *desc)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK) return 0;
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD_LOG2E) return 0;                  // a forward-only layout
    if (desc->variant == MVHMR_VARIANT_BRICK && !bwd_uses_brick(desc, p)) return 0;
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD && !bwd_uses_brick(desc, p) && !quad_to_channels_last_supported(p)) return 0;
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD && desc->variant == MVHMR_VARIANT_GATHER && !quad_to_channels_last_supported(p)) return 0;
    return 1;
}

size_t mvhmr_unproject_forward_workspace_bytes(const mvhmr_unproject_desc *desc)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK) return 0;
    if (desc->feat_layout == MVHMR_LAYOUT_BVHWC || desc->feat_layout == MVHMR_LAYOUT_QUAD_LOG2E) return 0;
    if (geometry_gated(desc, p)) return conv_bytes(p) + kGateBytes;   // one converted copy (either layout) + the gate counter
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD) return pick_variant(desc, p) == MVHMR_VARIANT_BRICK ? 0 : featT_bytes(p);
    return pick_variant(desc, p) == MVHMR_VARIANT_BRICK ? brick_workspace_bytes(p) : featT_bytes(p);
}

size_t mvhmr_unproject_backward_workspace_bytes(const mvhmr_unproject_desc *desc)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK || desc->feat_layout == MVHMR_LAYOUT_QUAD_LOG2E) return 0;
    if (geometry_gated_bwd(desc, p) && bwd_uses_brick(desc, p)) return conv_bytes(p) + bwd_mid_bytes(desc, p) + kGateBytes;
    if (bwd_uses_brick(desc, p)) return brick_workspace_bytes(p) + gradT_bytes(p);
    if (bwd_uses_plane(desc, p)) return (desc->feat_layout == MVHMR_LAYOUT_BVCHW ? brick_workspace_bytes(p) : 0) + align_up(plane_table_bytes(p));
    size_t need = desc->feat_layout != MVHMR_LAYOUT_BVHWC ? featT_bytes(p) : 0;
    if (!grad_in_place(desc, p)) need += gradT_bytes(p);
    return need;
}

static int forward_impl(const mvhmr_unproject_desc *desc, Problem &p, const void *features, const float *proj, const Coords &coords,
                        void *out, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    int rc;
    if (!features || !proj || !out) return fail(MVHMR_ERR_INVALID_ARGUMENT, "features / proj / out must be non-null");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    mvhmr_unproject_desc dq;
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD_LOG2E) {
        // the quad-planar copy times log2(e): only the wave-specialised softmax forward reads it (mvhmr_preferred_layout says when)
        if (!brick_fwd_prescales(p) || desc->variant == MVHMR_VARIANT_GATHER)
            return fail(MVHMR_ERR_UNSUPPORTED, "MVHMR_LAYOUT_QUAD_LOG2E feeds the softmax brick forward of 3 / 4 views with an fp32 volume only "
                                               "(ask mvhmr_preferred_layout)");
        dq = *desc;
        dq.feat_layout = MVHMR_LAYOUT_QUAD;
        dq.variant = MVHMR_VARIANT_BRICK;
        desc = &dq;
        p.feat_log2e = 1;
    }
    const int variant = pick_variant(desc, p);
    rc = variant_conflict(desc, p, variant);
    if (rc != MVHMR_OK) return rc;
    rc = check_ws(workspace, workspace_bytes, mvhmr_unproject_forward_workspace_bytes(desc));
    if (rc != MVHMR_OK) return rc;
    // planar input whose staged copy this call makes itself: scaled by log2(e) when the kernel that reads it wants that
    if (desc->feat_layout == MVHMR_LAYOUT_BVCHW && brick_fwd_prescales(p)) p.feat_log2e = 1;

    if (geometry_gated(desc, p)) {
        // both variants are launched; the device-side brick count lets exactly one of them (and its layout pass) run
        unsigned char *ws = static_cast<unsigned char *>(workspace);
        rc = arm_gate(p, ws + conv_bytes(p), proj, coords, brick_fwd_gate_geom(p), s);
        if (rc != MVHMR_OK) return rc;
        const bool quad = desc->feat_layout == MVHMR_LAYOUT_QUAD;
        if (!quad) {
            rc = launched(launch_to_quad_planar_t(features, ws, p, s), "layout pass");
            if (rc != MVHMR_OK) return rc;
        }
        rc = launched(quad ? launch_quad_to_channels_last(features, ws, p, s) : launch_to_channels_last(features, ws, p, s), "layout pass");
        if (rc != MVHMR_OK) return rc;
        rc = launched(launch_fwd_brick(quad ? features : ws, proj, coords, out, p, s), "brick forward");
        if (rc != MVHMR_OK) return rc;
        return launched(launch_fwd_gather(ws, proj, coords, out, p, s), "gather forward");
    }

    if (variant == MVHMR_VARIANT_BRICK) {
        const void *featK = features;
        if (desc->feat_layout == MVHMR_LAYOUT_BVCHW) {
            rc = launched(launch_to_quad_planar_t(features, workspace, p, s), "layout pass");
            if (rc != MVHMR_OK) return rc;
            featK = workspace;
        }
        return launched(launch_fwd_brick(featK, proj, coords, out, p, s), "brick forward");
    }

    const void *featT = features;
    if (desc->feat_layout != MVHMR_LAYOUT_BVHWC) {
        rc = launched(desc->feat_layout == MVHMR_LAYOUT_QUAD ? launch_quad_to_channels_last(features, workspace, p, s)
                                                             : launch_to_channels_last(features, workspace, p, s), "layout pass");
        if (rc != MVHMR_OK) return rc;
        featT = workspace;
    }
    return launched(launch_fwd_gather(featT, proj, coords, out, p, s), "gather forward");
}

int mvhmr_unproject_forward(const mvhmr_unproject_desc *desc, const void *features, const float *proj, const float *coords,
                            void *out, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    Problem p;
    int rc = check_desc(desc, &p);
    if (rc != MVHMR_OK) return rc;
    if (!coords) return fail(MVHMR_ERR_INVALID_ARGUMENT, "coords must be non-null");
    return forward_impl(desc, p, features, proj, coords_from_tensor(coords, p), out, workspace, workspace_bytes, hip_stream);
}

int mvhmr_unproject_forward_cuboid(const mvhmr_unproject_desc *desc, const void *features, const float *proj, const float *rot,
                                   const float *center, const double position[3], const double sides[3], void *out, void *workspace,
                                   size_t workspace_bytes, void *hip_stream)
{
    Problem p;
    int rc = check_desc(desc, &p);
    if (rc != MVHMR_OK) return rc;
    Coords cs;
    rc = coords_from_cuboid(rot, center, position, sides, p, &cs);
    if (rc != MVHMR_OK) return rc;
    return forward_impl(desc, p, features, proj, cs, out, workspace, workspace_bytes, hip_stream);
}

static int backward_impl(const mvhmr_unproject_desc *desc, Problem &p, const void *grad_out, const void *features, const float *proj,
                         const Coords &coords, void *grad_features, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    int rc;
    if (!grad_out || !features || !proj || !grad_features)
        return fail(MVHMR_ERR_INVALID_ARGUMENT, "grad_out / features / proj / grad_features must be non-null");
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD_LOG2E)
        return fail(MVHMR_ERR_UNSUPPORTED, "MVHMR_LAYOUT_QUAD_LOG2E is a forward-only layout: hand the backward the features as they are");
    if (desc->feat_layout == MVHMR_LAYOUT_QUAD && !bwd_uses_brick(desc, p) && !quad_to_channels_last_supported(p))
        return fail(MVHMR_ERR_UNSUPPORTED, "backward from quad-planar features of this shape needs the brick backward (fp32, 2 / 4 / 8 views)");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    rc = check_ws(workspace, workspace_bytes, mvhmr_unproject_backward_workspace_bytes(desc));
    if (rc != MVHMR_OK) return rc;

    unsigned char *ws = static_cast<unsigned char *>(workspace);
    if (desc->variant == MVHMR_VARIANT_BRICK && !bwd_uses_brick(desc, p))
        return fail(MVHMR_ERR_UNSUPPORTED, "the brick variant does not support this shape / dtype / layout");
    if (geometry_gated_bwd(desc, p) && bwd_uses_brick(desc, p)) {
        float *acc = reinterpret_cast<float *>(ws + conv_bytes(p));               // quad-planar or channels-last accumulator
        rc = arm_gate(p, ws + conv_bytes(p) + bwd_mid_bytes(desc, p), proj, coords, brick_bwd_gate_geom(p), s);
        if (rc != MVHMR_OK) return rc;
        const bool quad = desc->feat_layout == MVHMR_LAYOUT_QUAD;
        if (bwd_uses_plane(desc, p)) {
            // both sides read the column-major quad copy (the caller's own when the features came quad-planar).  Brick side: LDS windows,
            // float-atomic flush into `acc`, layout pass; gather side: the plane kernels, their scratch where `acc` would be
            if (!quad) {
                Problem pu = p;
                pu.gate_count = nullptr;                                         // not gated: either side needs the copy
                rc = launched(launch_to_quad_planar_t(features, ws, pu, s, true), "layout pass");
                if (rc != MVHMR_OK) return rc;
            }
            const void *featK = quad ? features : ws;
            hipError_t e = hipMemsetAsync(acc, 0, (size_t)p.B * p.V * p.H * p.W * p.C4 * sizeof(float), s);
            if (e != hipSuccess) return launched(e, "gradient clear");
            rc = launched(launch_bwd_brick(featK, grad_out, proj, coords, acc, p, s), "brick backward");
            if (rc != MVHMR_OK) return rc;
            rc = launched(launch_quad_grad_to_planar(acc, grad_features, p, s), "gradient layout pass");
            if (rc != MVHMR_OK) return rc;
            return launched(launch_bwd_plane(featK, grad_out, proj, coords, grad_features, acc, p, s), "plane backward");
        }
        // brick side: the column-major quad copy (the caller's own when the features came quad-planar); gather side: channels-last
        if (!quad) {
            rc = launched(launch_to_quad_planar_t(features, ws, p, s, true), "layout pass");
            if (rc != MVHMR_OK) return rc;
        }
        rc = launched(quad ? launch_quad_to_channels_last(features, ws, p, s) : launch_to_channels_last(features, ws, p, s), "layout pass");
        if (rc != MVHMR_OK) return rc;
        hipError_t e = hipMemsetAsync(acc, 0, (size_t)p.B * p.V * p.H * p.W * p.C4 * sizeof(float), s);
        if (e != hipSuccess) return launched(e, "gradient clear");
        rc = launched(launch_bwd_brick(quad ? features : ws, grad_out, proj, coords, acc, p, s), "brick backward");
        if (rc != MVHMR_OK) return rc;
        rc = launched(launch_bwd_gather(grad_out, ws, proj, coords, acc, p, s), "gather backward");
        if (rc != MVHMR_OK) return rc;
        rc = launched(launch_quad_grad_to_planar(acc, grad_features, p, s), "gradient layout pass");
        if (rc != MVHMR_OK) return rc;
        return launched(launch_grad_to_planar(acc, grad_features, p, s), "gradient layout pass");
    }
    if (bwd_uses_brick(desc, p)) {
        float *gradK = reinterpret_cast<float *>(ws + brick_workspace_bytes(p));
        const void *featK = features;                                            // quad-planar features: the copy as it is (column-major)
        if (desc->feat_layout != MVHMR_LAYOUT_QUAD) {
            rc = launched(launch_to_quad_planar_t(features, ws, p, s, true), "layout pass");
            if (rc != MVHMR_OK) return rc;
            featK = ws;
        }
        hipError_t e = hipMemsetAsync(gradK, 0, (size_t)p.B * p.V * p.H * p.W * p.C4 * sizeof(float), s);
        if (e != hipSuccess) return launched(e, "gradient clear");
        rc = launched(launch_bwd_brick(featK, grad_out, proj, coords, gradK, p, s), "brick backward");
        if (rc != MVHMR_OK) return rc;
        return launched(launch_quad_grad_to_planar(gradK, grad_features, p, s), "gradient layout pass");
    }
    if (bwd_uses_plane(desc, p)) {
        const void *featK = features;
        if (desc->feat_layout == MVHMR_LAYOUT_BVCHW) {
            rc = launched(launch_to_quad_planar_t(features, ws, p, s), "layout pass");
            if (rc != MVHMR_OK) return rc;
            featK = ws;
            ws += brick_workspace_bytes(p);
        }
        return launched(launch_bwd_plane(featK, grad_out, proj, coords, grad_features, ws, p, s), "plane backward");
    }
    const void *featT = features;
    if (desc->feat_layout != MVHMR_LAYOUT_BVHWC) {
        rc = launched(desc->feat_layout == MVHMR_LAYOUT_QUAD ? launch_quad_to_channels_last(features, ws, p, s) : launch_to_channels_last(features, ws, p, s),
                      "layout pass");
        if (rc != MVHMR_OK) return rc;
        featT = ws;
        ws += featT_bytes(p);
    }
    const bool in_place = grad_in_place(desc, p);
    float *gradT = in_place ? static_cast<float *>(grad_features) : reinterpret_cast<float *>(ws);
    const size_t gbytes = (size_t)p.B * p.V * p.H * p.W * p.C4 * sizeof(float);
    hipError_t e = hipMemsetAsync(gradT, 0, gbytes, s);
    if (e != hipSuccess) return launched(e, "gradient clear");
    rc = launched(launch_bwd_gather(grad_out, featT, proj, coords, gradT, p, s), "gather backward");
    if (rc != MVHMR_OK || in_place) return rc;
    if (desc->feat_layout != MVHMR_LAYOUT_BVHWC) return launched(launch_grad_to_planar(gradT, grad_features, p, s), "gradient layout pass");   // planar gradient for planar and quad-planar features alike
    return launched(launch_grad_cast(gradT, grad_features, p, s), "gradient cast");
}

int mvhmr_unproject_backward(const mvhmr_unproject_desc *desc, const void *grad_out, const void *features, const float *proj,
                             const float *coords, void *grad_features, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    Problem p;
    int rc = check_desc(desc, &p);
    if (rc != MVHMR_OK) return rc;
    if (!coords) return fail(MVHMR_ERR_INVALID_ARGUMENT, "coords must be non-null");
    return backward_impl(desc, p, grad_out, features, proj, coords_from_tensor(coords, p), grad_features, workspace, workspace_bytes, hip_stream);
}

int mvhmr_unproject_backward_cuboid(const mvhmr_unproject_desc *desc, const void *grad_out, const void *features, const float *proj,
                                    const float *rot, const float *center, const double position[3], const double sides[3],
                                    void *grad_features, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    Problem p;
    int rc = check_desc(desc, &p);
    if (rc != MVHMR_OK) return rc;
    Coords cs;
    rc = coords_from_cuboid(rot, center, position, sides, p, &cs);
    if (rc != MVHMR_OK) return rc;
    return backward_impl(desc, p, grad_out, features, proj, cs, grad_features, workspace, workspace_bytes, hip_stream);
}

int mvhmr_preferred_layout(const mvhmr_unproject_desc *desc)
{
    Problem p;
    if (!desc) return -1;
    mvhmr_unproject_desc d = *desc;
    d.feat_layout = MVHMR_LAYOUT_BVCHW;
    if (check_desc(&d, &p) != MVHMR_OK) return -1;
    const int variant = pick_variant(&d, p);
    if (variant_conflict(&d, p, variant) != MVHMR_OK) return -1;
    if (variant != MVHMR_VARIANT_BRICK) return MVHMR_LAYOUT_BVHWC;
    return brick_fwd_prescales(p) ? MVHMR_LAYOUT_QUAD_LOG2E : MVHMR_LAYOUT_QUAD;
}

size_t mvhmr_feature_layout_bytes(const mvhmr_unproject_desc *desc, int dst_layout)
{
    Problem p;
    if (!desc) return 0;
    mvhmr_unproject_desc d = *desc;
    d.feat_layout = MVHMR_LAYOUT_BVCHW;
    if (check_desc(&d, &p) != MVHMR_OK) return 0;
    if (dst_layout == MVHMR_LAYOUT_BVHWC) return featT_bytes(p);
    if (dst_layout == MVHMR_LAYOUT_QUAD || dst_layout == MVHMR_LAYOUT_QUAD_LOG2E) return brick_workspace_bytes(p);   // always fp32, whatever the storage type; (C + 3) / 4 quads per view
    return 0;
}

int mvhmr_convert_features(const mvhmr_unproject_desc *desc, const void *features, int dst_layout, void *dst, void *hip_stream)
{
    Problem p;
    if (!desc) return fail(MVHMR_ERR_INVALID_ARGUMENT, "descriptor is null");
    mvhmr_unproject_desc d = *desc;
    const bool from_channels_last = desc->feat_layout == MVHMR_LAYOUT_BVHWC;      // r04: a channels-last source (to MVHMR_LAYOUT_QUAD only)
    d.feat_layout = MVHMR_LAYOUT_BVCHW;
    int rc = check_desc(&d, &p);
    if (rc != MVHMR_OK) return rc;
    if (!features || !dst) return fail(MVHMR_ERR_INVALID_ARGUMENT, "features / dst must be non-null");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (from_channels_last) {
        if (dst_layout != MVHMR_LAYOUT_QUAD) return fail(MVHMR_ERR_UNSUPPORTED, "channels-last features convert to MVHMR_LAYOUT_QUAD only");
        p.gate_count = nullptr;
        return launched(launch_channels_last_to_quad_planar_t(features, dst, p, s), "layout pass");
    }
    if (dst_layout == MVHMR_LAYOUT_BVHWC) return launched(launch_to_channels_last(features, dst, p, s), "layout pass");
    if (dst_layout == MVHMR_LAYOUT_QUAD) return launched(launch_to_quad_planar_t(features, dst, p, s), "layout pass");
    if (dst_layout == MVHMR_LAYOUT_QUAD_LOG2E) {
        p.feat_log2e = 1;
        return launched(launch_to_quad_planar_t(features, dst, p, s), "layout pass");
    }
    return fail(MVHMR_ERR_INVALID_ARGUMENT, "unknown destination layout %d", dst_layout);
}

int mvhmr_conv1x1_to_quad(const float *x, const float *weight, const float *bias, void *dst, int32_t n_maps, int32_t c_in, int32_t c_out,
                          int32_t feat_h, int32_t feat_w, void *hip_stream)
{
    if (!x || !weight || !dst) return fail(MVHMR_ERR_INVALID_ARGUMENT, "x / weight / dst must be non-null");
    if (n_maps < 1 || c_in < 1 || c_out < 1 || feat_h < 1 || feat_w < 1) return fail(MVHMR_ERR_INVALID_ARGUMENT, "every dimension must be >= 1");
    if (!conv1x1_quad_supported(c_in, c_out, feat_h, feat_w))
        return fail(MVHMR_ERR_UNSUPPORTED, "fused 1x1 conv needs C_in %% 16 == 0, C_out %% 128 == 0, Hf %% 4 == 0, Wf %% 32 == 0 (got %d -> %d, %dx%d)",
                    c_in, c_out, feat_h, feat_w);
    return launched(launch_conv1x1_quad(x, weight, bias, dst, n_maps, c_in, c_out, feat_h, feat_w, static_cast<hipStream_t>(hip_stream)),
                    "fused 1x1 conv");
}

int mvhmr_conv1x1_to_quad_supported(int32_t c_in, int32_t c_out, int32_t feat_h, int32_t feat_w)
{
    return conv1x1_quad_supported(c_in, c_out, feat_h, feat_w) ? 1 : 0;
}

int mvhmr_conv1x1_planar(const float *x, const float *weight, const float *bias, float *dst, int32_t n_maps, int32_t c_in, int32_t c_out,
                         int32_t pixels, void *hip_stream)
{
    if (!x || !weight || !dst) return fail(MVHMR_ERR_INVALID_ARGUMENT, "null pointer");
    if (n_maps <= 0 || c_in <= 0 || c_out <= 0 || pixels <= 0) return fail(MVHMR_ERR_INVALID_ARGUMENT, "non-positive extent");
    if (!conv1x1_planar_supported(c_in, c_out, pixels))
        return fail(MVHMR_ERR_UNSUPPORTED, "planar 1x1 conv needs C_in %% 16 == 0, C_out %% 128 == 0, pixels %% 128 == 0 (got %d -> %d, %d)", c_in, c_out, pixels);
    return launched(launch_conv1x1_planar(x, weight, bias, dst, n_maps, c_in, c_out, pixels, static_cast<hipStream_t>(hip_stream)), "planar 1x1 conv");
}

int mvhmr_conv1x1_planar_supported(int32_t c_in, int32_t c_out, int32_t pixels)
{
    return conv1x1_planar_supported(c_in, c_out, pixels) ? 1 : 0;
}

int mvhmr_conv1x1_wgrad(const float *grad_y, const float *x, float *grad_weight, float *grad_bias, int32_t n_maps, int32_t c_in,
                        int32_t c_out, int32_t pixels, void *hip_stream)
{
    if (!grad_y || !x || !grad_weight) return fail(MVHMR_ERR_INVALID_ARGUMENT, "null pointer");
    if (n_maps <= 0 || c_in <= 0 || c_out <= 0 || pixels <= 0) return fail(MVHMR_ERR_INVALID_ARGUMENT, "non-positive extent");
    if (!conv1x1_wgrad_supported(c_in, c_out, pixels))
        return fail(MVHMR_ERR_UNSUPPORTED, "1x1 conv weight gradient needs C_in %% 128 == 0, C_out %% 128 == 0, pixels %% 32 == 0 (got %d -> %d, %d)", c_in, c_out, pixels);
    return launched(launch_conv1x1_wgrad(grad_y, x, grad_weight, grad_bias, n_maps, c_in, c_out, pixels, static_cast<hipStream_t>(hip_stream)),
                    "1x1 conv weight gradient");
}

int mvhmr_conv1x1_wgrad_supported(int32_t c_in, int32_t c_out, int32_t pixels)
{
    return conv1x1_wgrad_supported(c_in, c_out, pixels) ? 1 : 0;
}

int mvhmr_unproject_query_variant_cuboid(const mvhmr_unproject_desc *desc, const float *proj, const float *rot, const float *center,
                                         const double position[3], const double sides[3], void *hip_stream)
{
    Problem p;
    if (check_desc(desc, &p) != MVHMR_OK) return -1;
    const int variant = pick_variant(desc, p);
    if (variant_conflict(desc, p, variant) != MVHMR_OK) return -1;
    if (desc->variant != MVHMR_VARIANT_AUTO || !brick_fwd_preferred(p)) return variant;
    Coords cs;
    if (!proj || coords_from_cuboid(rot, center, position, sides, p, &cs) != MVHMR_OK) { fail(MVHMR_ERR_INVALID_ARGUMENT, "null pointer"); return -1; }
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int *count = nullptr, host = 0;
    if (hipMalloc(&count, sizeof(int)) != hipSuccess) { fail(MVHMR_ERR_LAUNCH, "query: allocation failed"); return -1; }
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
    const GateGeom g = brick_fwd_gate_geom(p);
    if (e == hipSuccess) e = launch_brick_gate(proj, cs, count, g, p, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&host, count, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(count);
    if (e != hipSuccess) { fail(MVHMR_ERR_LAUNCH, "query: %s", hipGetErrorString(e)); return -1; }
    return host <= brick_count(p, g) / 8 ? MVHMR_VARIANT_BRICK : MVHMR_VARIANT_GATHER;
}

int mvhmr_triangulate_dlt(const float *proj, const float *points, float *out, int32_t batch, int32_t views, int32_t points_per_sample,
                          void *hip_stream)
{
    if (!proj || !points || !out) return fail(MVHMR_ERR_INVALID_ARGUMENT, "null pointer");
    if (batch < 1 || views < 2) return fail(MVHMR_ERR_INVALID_ARGUMENT, "batch must be >= 1 and views >= 2 (got %d, %d)", batch, views);
    return launched(launch_triangulate_dlt(proj, points, nullptr, out, batch, views, points_per_sample ? 1 : 0, 0, static_cast<hipStream_t>(hip_stream)),
                    "DLT triangulation");
}

int mvhmr_triangulate_dlt_weighted(const float *proj, const float *points, const float *confidences, float *out, int32_t batch, int32_t views,
                                   int32_t points_per_sample, int32_t confidences_per_sample, void *hip_stream)
{
    if (!proj || !points || !confidences || !out) return fail(MVHMR_ERR_INVALID_ARGUMENT, "null pointer");
    if (batch < 1 || views < 2) return fail(MVHMR_ERR_INVALID_ARGUMENT, "batch must be >= 1 and views >= 2 (got %d, %d)", batch, views);
    return launched(launch_triangulate_dlt(proj, points, confidences, out, batch, views, points_per_sample ? 1 : 0, confidences_per_sample ? 1 : 0,
                                           static_cast<hipStream_t>(hip_stream)), "weighted DLT triangulation");
}

int mvhmr_build_coord_volumes(float *coords, const float *rot, const float *center, int32_t batch, int32_t volume_size,
                              const double position[3], const double sides[3], void *hip_stream)
{
    if (!coords || !rot || !center || !position || !sides) return fail(MVHMR_ERR_INVALID_ARGUMENT, "null pointer");
    if (batch < 1 || volume_size < 1) return fail(MVHMR_ERR_INVALID_ARGUMENT, "batch and volume_size must be >= 1");
    return launched(launch_build_coords(coords, rot, center, batch, volume_size, position, sides, static_cast<hipStream_t>(hip_stream)),
                    "coord volume build");
}

}  // extern "C"
