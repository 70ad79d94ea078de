// PyTorch-ROCm C++ extension over the C ABI (include/mvhmr_unproject.h): the native implementation of the custom ops
// mvhmr::unprojection_native / mvhmr::unprojection_backward_native, which multiviewhmr_amd/aggregation.py dispatches to from its
// torch.library ops mvhmr::unprojection / _backward (reference boundary: unprojection(), models/aggregation.py:20-87, and the autograd
// graph through it).  What runs here per call: descriptor, output / gradient tensor and workspace from the caching allocator, the
// current HIP stream, one C-ABI call -- no Python, no ctypes marshalling.  Host code only: the kernels live in libmvhmr_unproject.so.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include "mvhmr_unproject.h"

namespace {

mvhmr_unproject_desc make_desc(int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, const at::Tensor &coords, int64_t method,
                               int64_t feat_dtype, int64_t out_dtype, int64_t layout, int64_t variant)
{
    TORCH_CHECK(coords.dim() == 5 && coords.size(4) == 3, "coord_volumes must be (B, X, Y, Z, 3)");
    mvhmr_unproject_desc d;
    d.abi_version = MVHMR_ABI_VERSION;
    d.batch = (int32_t)B; d.views = (int32_t)V; d.channels = (int32_t)C; d.feat_h = (int32_t)H; d.feat_w = (int32_t)W;
    d.vol_x = (int32_t)coords.size(1); d.vol_y = (int32_t)coords.size(2); d.vol_z = (int32_t)coords.size(3);
    d.method = (int32_t)method; d.feat_dtype = (int32_t)feat_dtype; d.out_dtype = (int32_t)out_dtype;
    d.feat_layout = (int32_t)layout; d.variant = (int32_t)variant;
    return d;
}

at::ScalarType scalar_of(int64_t code)
{
    return code == MVHMR_F16 ? at::kHalf : code == MVHMR_BF16 ? at::kBFloat16 : at::kFloat;
}

void check(int status)
{
    TORCH_CHECK(status == MVHMR_OK, "mvhmr_unproject: ", mvhmr_last_error());
}

// The C ABI trusts its descriptor (plain pointers carry no sizes): every tensor is checked against it HERE, before anything is launched
void check_sizes(const at::Tensor &features, const at::Tensor &proj, const at::Tensor &coords, int64_t B, int64_t V, int64_t C, int64_t H,
                 int64_t W, int64_t feat_dtype, int64_t layout)
{
    TORCH_CHECK(B >= 1 && V >= 1 && C >= 1 && H >= 1 && W >= 1, "mvhmr_unproject: every dimension must be >= 1");
    TORCH_CHECK(feat_dtype == MVHMR_F32 || feat_dtype == MVHMR_F16, "mvhmr_unproject: features are fp32 or fp16");
    const int64_t quad = layout == MVHMR_LAYOUT_QUAD || layout == MVHMR_LAYOUT_QUAD_LOG2E;
    const int64_t need = B * V * ((C + 3) / 4 * 4) * H * W * (quad ? 4 : feat_dtype == MVHMR_F16 ? 2 : 4);
    const int64_t have = features.numel() * (int64_t)features.element_size();
    TORCH_CHECK(quad || layout == MVHMR_LAYOUT_BVHWC ? have >= need : have == B * V * C * H * W * (int64_t)features.element_size(),
                "mvhmr_unproject: features hold ", have, " bytes, the descriptor (", B, ", ", V, ", ", C, ", ", H, ", ", W, ") needs ", need);
    TORCH_CHECK(quad || features.element_size() == (feat_dtype == MVHMR_F16 ? 2 : 4), "mvhmr_unproject: feature dtype and descriptor disagree");
    TORCH_CHECK(proj.scalar_type() == at::kFloat && proj.numel() == B * V * 12, "mvhmr_unproject: proj_matricies must be fp32 (B, V, 3, 4)");
    TORCH_CHECK(coords.scalar_type() == at::kFloat && coords.dim() == 5 && coords.size(0) == B && coords.size(4) == 3,
                "mvhmr_unproject: coord_volumes must be fp32 (B, X, Y, Z, 3)");
}

// features: the tensor the library reads (planar, channels-last or the quad-planar byte buffer); B..W: the logical feature shape
at::Tensor unprojection_native(const at::Tensor &features, const at::Tensor &proj, const at::Tensor &coords, int64_t B, int64_t V, int64_t C,
                               int64_t H, int64_t W, int64_t method, int64_t feat_dtype, int64_t out_dtype, int64_t layout, int64_t variant)
{
    TORCH_CHECK(features.is_cuda() && proj.is_cuda() && coords.is_cuda(), "unprojection runs only on a HIP device");
    TORCH_CHECK(features.is_contiguous() && proj.is_contiguous() && coords.is_contiguous(), "contiguous tensors expected");
    check_sizes(features, proj, coords, B, V, C, H, W, feat_dtype, layout);
    c10::DeviceGuard guard(features.device());
    const mvhmr_unproject_desc d = make_desc(B, V, C, H, W, coords, method, feat_dtype, out_dtype, layout, variant);
    at::Tensor out = at::empty({B, C, coords.size(1), coords.size(2), coords.size(3)}, features.options().dtype(scalar_of(out_dtype)));
    const size_t need = mvhmr_unproject_forward_workspace_bytes(&d);
    at::Tensor ws = at::empty({(int64_t)need}, features.options().dtype(at::kByte));
    check(mvhmr_unproject_forward(&d, features.data_ptr(), proj.data_ptr<float>(), coords.data_ptr<float>(), out.data_ptr(),
                                  need ? ws.data_ptr() : nullptr, need, c10::hip::getCurrentHIPStream(features.device().index()).stream()));
    return out;
}

// gradient w.r.t. the features, in the feature dtype; planar (B,V,C,H,W) -- or channels-last strides when layout is BVHWC
at::Tensor unprojection_backward_native(const at::Tensor &grad_out, const at::Tensor &features, const at::Tensor &proj, const at::Tensor &coords,
                                        int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t method, int64_t feat_dtype,
                                        int64_t out_dtype, int64_t layout, int64_t variant)
{
    TORCH_CHECK(grad_out.is_cuda() && features.is_cuda(), "unprojection runs only on a HIP device");
    TORCH_CHECK(grad_out.is_contiguous() && features.is_contiguous(), "contiguous tensors expected");
    check_sizes(features, proj, coords, B, V, C, H, W, feat_dtype, layout);
    TORCH_CHECK(grad_out.scalar_type() == scalar_of(out_dtype) && grad_out.numel() == B * C * coords.size(1) * coords.size(2) * coords.size(3),
                "mvhmr_unproject: grad_out must be (B, C, X, Y, Z) in the volume's dtype");
    c10::DeviceGuard guard(features.device());
    const mvhmr_unproject_desc d = make_desc(B, V, C, H, W, coords, method, feat_dtype, out_dtype, layout, variant);
    const auto opts = features.options().dtype(scalar_of(feat_dtype));
    at::Tensor grad = layout == MVHMR_LAYOUT_BVHWC ? at::empty({B, V, H, W, C}, opts) : at::empty({B, V, C, H, W}, opts);
    const size_t need = mvhmr_unproject_backward_workspace_bytes(&d);
    at::Tensor ws = at::empty({(int64_t)need}, features.options().dtype(at::kByte));
    check(mvhmr_unproject_backward(&d, grad_out.data_ptr(), features.data_ptr(), proj.data_ptr<float>(), coords.data_ptr<float>(), grad.data_ptr(),
                                   need ? ws.data_ptr() : nullptr, need, c10::hip::getCurrentHIPStream(features.device().index()).stream()));
    return layout == MVHMR_LAYOUT_BVHWC ? grad.permute({0, 1, 4, 2, 3}) : grad;
}

}  // namespace

TORCH_LIBRARY(mvhmr_native, m)
{
    m.def("unprojection(Tensor features, Tensor proj, Tensor coords, int B, int V, int C, int H, int W, int method, int feat_dtype, int out_dtype, "
          "int layout, int variant) -> Tensor");
    m.def("unprojection_backward(Tensor grad_out, Tensor features, Tensor proj, Tensor coords, int B, int V, int C, int H, int W, int method, "
          "int feat_dtype, int out_dtype, int layout, int variant) -> Tensor");
    m.def("abi_version() -> int");
}

TORCH_LIBRARY_IMPL(mvhmr_native, CUDA, m)
{
    m.impl("unprojection", &unprojection_native);
    m.impl("unprojection_backward", &unprojection_backward_native);
}

TORCH_LIBRARY_IMPL(mvhmr_native, CompositeExplicitAutograd, m)
{
    m.impl("abi_version", []() -> int64_t { return mvhmr_abi_version(); });
}
