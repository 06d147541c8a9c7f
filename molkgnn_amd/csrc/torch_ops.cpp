// PyTorch dispatcher registration of the kernel convolution (TORCH_LIBRARY): torch.ops.molkgnn.kernelsetconv_forward /
// kernelsetconv_backward / backward_join.  A thin shim over the C ABI (include/molkgnn_hip.h): tensors in, the ABI's
// structs of device pointers filled here, the caller's current HIP stream passed on; no kernel of its own.  It exists
// for callers that want the path as registered operators (the reference is pure PyTorch, kernels.py:610-751: an eager
// model pays Python + ctypes marshalling per call; here one dispatcher call), and builds into its own shared library
// (libmolkgnn_torch.so, linked against libmolkgnn_hip.so) so that the C-ABI library itself stays free of torch types.
//
// Argument layout (flat lists, degree 1..4 in order):
//   params   28 tensors: per degree x_center [L, F], x_support [L, d, F], edge_attr_support [L, d, E], p_support [L, d, 3],
//            support_attr_sc_weight, center_attr_sc_weight, edge_attr_support_sc_weight (0-d)     (kernels.py:50-84)
//   buckets  24 tensors: per degree selected_index, nei_index (int64), nei_edge_attr, p_focal, nei_p, nei_edge_unit
//            (an empty tensor = absent)                                                         (wrapper.py:596-635)
//   counts   4 ints: atoms per bucket
//   saved    8 tensors: per degree pair_state [N_d, L_d, 4], chirality [N_d, L_d] int8 (empty = not kept)
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/molkgnn_hip.h"

namespace {

const float* fptr(const at::Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }
template <typename T> T* ptr_or_null(const at::Tensor& t) { return (t.defined() && t.numel() > 0) ? (T*)t.data_ptr() : nullptr; }

void check_device(const at::Tensor& t, const char* what) {
    TORCH_CHECK(!t.defined() || t.numel() == 0 || t.is_cuda(), "molkgnn: ", what, " must be a GPU tensor (there is no CPU path)");
}

void fill(const at::TensorList params, const at::TensorList buckets, at::IntArrayRef counts, mkgnn_kernel_bank banks[4],
          mkgnn_degree_bucket bk[4]) {
    TORCH_CHECK(params.size() == 28, "molkgnn: 28 parameter tensors (7 per degree), got ", params.size());
    TORCH_CHECK(buckets.size() == 24, "molkgnn: 24 bucket tensors (6 per degree), got ", buckets.size());
    TORCH_CHECK(counts.size() == 4, "molkgnn: 4 bucket sizes");
    for (int i = 0; i < 4; ++i) {
        const at::Tensor* p = &params[7 * i];
        for (int k = 0; k < 7; ++k) {
            check_device(p[k], "a kernel parameter");
            TORCH_CHECK(p[k].scalar_type() == at::kFloat && p[k].is_contiguous(), "molkgnn: kernel parameters are contiguous fp32");
        }
        mkgnn_kernel_bank& b = banks[i];
        b.num_kernels = (int32_t)p[0].size(0);
        b.reserved = 0;
        b.x_center = fptr(p[0]); b.x_support = fptr(p[1]); b.edge_attr_support = fptr(p[2]);
        b.p_support = (p[3].dim() == 3 && p[3].size(2) == 3) ? fptr(p[3]) : nullptr;
        b.support_attr_sc_weight = p[4].data_ptr<float>();
        b.center_attr_sc_weight = p[5].data_ptr<float>();
        b.edge_attr_support_sc_weight = p[6].data_ptr<float>();
        const at::Tensor* q = &buckets[6 * i];
        for (int k = 0; k < 6; ++k) check_device(q[k], "a bucket tensor");
        mkgnn_degree_bucket& g = bk[i];
        g.count = counts[i];
        g.selected_index = ptr_or_null<const int64_t>(q[0]);
        g.nei_index = ptr_or_null<const int64_t>(q[1]);
        g.nei_edge_attr = fptr(q[2]); g.p_focal = fptr(q[3]); g.nei_p = fptr(q[4]); g.nei_edge_unit = fptr(q[5]);
    }
}

void fill_saved(const at::TensorList saved, mkgnn_saved sv[4]) {
    TORCH_CHECK(saved.size() == 8, "molkgnn: 8 saved-state tensors (2 per degree), got ", saved.size());
    for (int i = 0; i < 4; ++i) {
        check_device(saved[2 * i], "pair_state"); check_device(saved[2 * i + 1], "chirality");
        sv[i].pair_state = ptr_or_null<float>(saved[2 * i]);
        sv[i].chirality = ptr_or_null<int8_t>(saved[2 * i + 1]);
    }
}

void* current_stream(const at::Tensor& x) { return (void*)c10::hip::getCurrentHIPStream(x.get_device()).stream(); }

void kernelsetconv_forward(const at::Tensor& x, const at::Tensor& inv_norm, at::TensorList params, at::TensorList buckets,
                           at::IntArrayRef counts, int64_t E, bool is_last_layer, at::Tensor& out, at::TensorList saved,
                           at::Tensor& workspace, int64_t variant) {
    check_device(x, "x"); check_device(out, "out"); check_device(workspace, "workspace");
    TORCH_CHECK(x.dim() == 2 && x.stride(1) == 1 && out.dim() == 2 && out.stride(1) == 1, "molkgnn: x and out are row-major matrices");
    mkgnn_kernel_bank banks[4]; mkgnn_degree_bucket bk[4]; mkgnn_saved sv[4];
    fill(params, buckets, counts, banks, bk);
    fill_saved(saved, sv);
    const int rc = mkgnn_kernelsetconv_forward(banks, bk, x.data_ptr<float>(), x.stride(0), inv_norm.data_ptr<float>(), x.size(0),
                                               (int32_t)x.size(1), (int32_t)E, is_last_layer ? 1 : 0, out.data_ptr<float>(),
                                               out.stride(0), sv, workspace.data_ptr(), (size_t)workspace.numel() * workspace.element_size(),
                                               (int32_t)variant, current_stream(x));
    TORCH_CHECK(rc == 0, "mkgnn_kernelsetconv_forward: ", mkgnn_last_error());
}

void kernelsetconv_backward(const at::Tensor& x, const at::Tensor& inv_norm, at::TensorList params, at::TensorList buckets,
                            at::IntArrayRef counts, int64_t E, bool is_last_layer, const at::Tensor& grad_out, at::TensorList saved,
                            const at::Tensor& scatter_rowptr, const at::Tensor& scatter_rows, const c10::optional<at::Tensor>& grad_x,
                            at::TensorList grads, at::Tensor& workspace, bool workspace_from_forward, int64_t variant) {
    check_device(x, "x"); check_device(grad_out, "grad_out"); check_device(workspace, "workspace");
    TORCH_CHECK(grads.size() == 16, "molkgnn: 16 gradient tensors (x_center, x_support, edge_attr_support, 3 score weights per degree)");
    mkgnn_kernel_bank banks[4]; mkgnn_degree_bucket bk[4]; mkgnn_saved sv[4]; mkgnn_kernel_bank_grad gr[4];
    fill(params, buckets, counts, banks, bk);
    fill_saved(saved, sv);
    for (int i = 0; i < 4; ++i) {
        const at::Tensor* g = &grads[4 * i];
        for (int k = 0; k < 4; ++k) check_device(g[k], "a gradient buffer");
        gr[i].x_center = ptr_or_null<float>(g[0]); gr[i].x_support = ptr_or_null<float>(g[1]);
        gr[i].edge_attr_support = ptr_or_null<float>(g[2]);
        TORCH_CHECK(g[3].numel() == 3 && g[3].is_contiguous(), "molkgnn: the score-weight gradients of a degree are one 3-element tensor");
        float* th = g[3].data_ptr<float>();
        gr[i].support_attr_sc_weight = th; gr[i].center_attr_sc_weight = th + 1; gr[i].edge_attr_support_sc_weight = th + 2;
    }
    float* gx = nullptr;
    int64_t gxs = 0;
    if (grad_x.has_value() && grad_x->defined()) { gx = grad_x->data_ptr<float>(); gxs = grad_x->stride(0); }
    const int rc = mkgnn_kernelsetconv_backward(banks, bk, x.data_ptr<float>(), x.stride(0), inv_norm.data_ptr<float>(), x.size(0),
                                                (int32_t)x.size(1), (int32_t)E, is_last_layer ? 1 : 0, grad_out.data_ptr<float>(),
                                                grad_out.stride(0), sv, scatter_rowptr.data_ptr<int32_t>(),
                                                ptr_or_null<const int32_t>(scatter_rows), gx, gxs, gr, workspace.data_ptr(),
                                                (size_t)workspace.numel() * workspace.element_size(), workspace_from_forward ? 1 : 0,
                                                (int32_t)variant, current_stream(x));
    TORCH_CHECK(rc == 0, "mkgnn_kernelsetconv_backward: ", mkgnn_last_error());
}

void backward_join(const at::Tensor& any_gpu_tensor) {
    const int rc = mkgnn_backward_join(current_stream(any_gpu_tensor));
    TORCH_CHECK(rc == 0, "mkgnn_backward_join: ", mkgnn_last_error());
}

int64_t abi_version() { return mkgnn_abi_version(); }

}  // namespace

TORCH_LIBRARY(molkgnn, m) {
    m.def("kernelsetconv_forward(Tensor x, Tensor inv_norm, Tensor[] params, Tensor[] buckets, int[] counts, int E, "
          "bool is_last_layer, Tensor(a!) out, Tensor(b!)[] saved, Tensor(c!) workspace, int variant) -> ()");
    m.def("kernelsetconv_backward(Tensor x, Tensor inv_norm, Tensor[] params, Tensor[] buckets, int[] counts, int E, "
          "bool is_last_layer, Tensor grad_out, Tensor[] saved, Tensor scatter_rowptr, Tensor scatter_rows, Tensor(a!)? grad_x, "
          "Tensor(b!)[] grads, Tensor(c!) workspace, bool workspace_from_forward, int variant) -> ()");
    m.def("backward_join(Tensor any_gpu_tensor) -> ()");
    m.def("abi_version() -> int");
}

TORCH_LIBRARY_IMPL(molkgnn, CUDA, m) {
    m.impl("kernelsetconv_forward", kernelsetconv_forward);
    m.impl("kernelsetconv_backward", kernelsetconv_backward);
    m.impl("backward_join", backward_join);
}

TORCH_LIBRARY_IMPL(molkgnn, CompositeExplicitAutograd, m) { m.impl("abi_version", abi_version); }
