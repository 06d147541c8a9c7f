// AdamW step of the whole model in one launch (gfx950).
//
// The reference trains with torch.optim.AdamW over two parameter groups (model.py:368-385: the kernel banks exempt
// from weight decay).  MolKGNN has ~80 small trainable tensors (132 k floats in all): PyTorch's fused multi-tensor
// path needs two launches per group plus one per group for the step counters -- five launch-bound kernels, ~40 us
// of a 1.3 ms training step, for 2 MB of traffic.  Here the tensor table travels in the kernel arguments (pointers
// are baked into a captured graph exactly like PyTorch's), one block per 1024 elements, each with its own copy of the
// tensor's step counter (round 2: a one-block kernel ahead of it advanced the counters -- 10 us for the pair; a "last block
// done" counter inside the update kernel was measured at 27 us for the two device-scope fences it needs).
//
// Arithmetic: the update of torch's fused kernel (fused_adam_utils.cuh, ADAMW mode), in the same order:
//   p -= lr * wd * p;  m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "kgnn_launch.h"
#include "../../include/molkgnn_hip.h"

namespace mkgnn {

constexpr int ADAM_MAX_TENSORS = 80;      // per launch: 80 * 40 B of kernel arguments
constexpr int ADAM_MAX_GROUPS = 4;
constexpr int ADAM_CHUNK = 1024;          // elements per block

struct AdamTensor { float* p; const float* g; float* state; const float* active; int32_t n; int32_t group; };
struct AdamGroup { const float* lr_ptr; float lr, beta1, beta2, eps, wd; int32_t maximize; float gscale; };
struct AdamArgs {
    AdamTensor t[ADAM_MAX_TENSORS];
    AdamGroup grp[ADAM_MAX_GROUPS];
    int32_t blk_start[ADAM_MAX_TENSORS + 1];
    int32_t nt;
};

// state of a tensor: [exp_avg | exp_avg_sq | step | 2 reserved | one step counter per block of the tensor].  Every block
// advances ITS OWN copy of the step count and derives the bias corrections from it (two double-precision pow per block), so
// no block reads a counter another block writes: the separate one-block counting kernel of round 2 (a dependent launch,
// 4 us of every step) is gone.  Block 0 also writes the canonical `step` the caller's state_dict reads.
__global__ void __launch_bounds__(256) adamw_step_kernel(AdamArgs a) {
    __shared__ float bc[2];
    // block -> tensor: the last ti with blk_start[ti] <= blockIdx.x
    int lo = 0, hi = a.nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamTensor T = a.t[lo];
    if (T.active && *T.active == 0.f) return;                // (block-uniform) no gradient anywhere this step: the step count stands still
    const AdamGroup G = a.grp[T.group];
    const float lr = G.lr_ptr ? *G.lr_ptr : G.lr;
    float* m = T.state;
    float* v = T.state + T.n;
    const int blk = (int)blockIdx.x - a.blk_start[lo];
    const int base = blk * ADAM_CHUNK;
    float g[ADAM_CHUNK / 256], p[ADAM_CHUNK / 256], mi[ADAM_CHUNK / 256], vi[ADAM_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {          // all loads first (clamped), then the arithmetic
        const int i = base + 256 * k + (int)threadIdx.x;
        const int ic = i < T.n ? i : T.n - 1;
        g[k] = T.g[ic]; p[k] = T.p[ic]; mi[k] = m[ic]; vi[k] = v[ic];
    }
    if (threadIdx.x == 0) {
        float* tail = T.state + 2 * (size_t)T.n;
        const double t = (double)tail[3 + blk] + 1.0;
        tail[3 + blk] = (float)t;
        if (blk == 0) tail[0] = (float)t;
        bc[0] = (float)(1.0 - pow((double)G.beta1, t));
        bc[1] = sqrtf((float)(1.0 - pow((double)G.beta2, t)));
    }
    __syncthreads();
    const float bc1 = bc[0], bc2_sqrt = bc[1];
    const float step_size = lr / bc1;
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int i = base + 256 * k + (int)threadIdx.x;
        if (i < T.n) {
            const float gg = (G.maximize ? -g[k] : g[k]) * G.gscale;
            float pp = p[k];
            pp -= lr * G.wd * pp;
            const float mm = fmaf(1.f - G.beta1, gg - mi[k], mi[k]);
            const float vv = G.beta2 * vi[k] + (1.f - G.beta2) * gg * gg;
            const float denom = sqrtf(vv) / bc2_sqrt + G.eps;
            pp -= step_size * mm / denom;
            T.p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

// Many small tensors into their slots of one flat buffer (or back) in ONE launch: the data-parallel step's gradients into the
// buffer RCCL sums (dp.FlatGradAllReduce).  torch._foreach_copy_ over the model's ~50 gradients is two multi-tensor launches of
// 12 us each at the end of every step's backward; this is one of launch-floor length (0.5 MB).
constexpr int COPY_MAX_ITEMS = 120;
struct CopyItem { float* dst; const float* src; int32_t n; int32_t pad; };
struct CopyArgs { CopyItem t[COPY_MAX_ITEMS]; int32_t blk_start[COPY_MAX_ITEMS + 1]; int32_t nt; };
__global__ void __launch_bounds__(256) flat_copy_kernel(CopyArgs a) {
    int lo = 0, hi = a.nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.blk_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const CopyItem T = a.t[lo];
    const int base = ((int)blockIdx.x - a.blk_start[lo]) * ADAM_CHUNK;
    float v[ADAM_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int i = base + 256 * k + (int)threadIdx.x;
        v[k] = T.src[i < T.n ? i : T.n - 1];
    }
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int i = base + 256 * k + (int)threadIdx.x;
        if (i < T.n) T.dst[i] = v[k];
    }
}

}  // namespace mkgnn

using namespace mkgnn;

extern "C" int mkgnn_flat_copy(const mkgnn_copy_item* items, int32_t n_items, void* stream) {
    if (n_items < 0) return api_fail("mkgnn_flat_copy: %d items", n_items);
    if (n_items == 0) return 0;
    if (!items) return api_fail("mkgnn_flat_copy: null pointer");
    hipStream_t st = (hipStream_t)stream;
    CopyArgs a{};
    for (int32_t first = 0; first < n_items; first += COPY_MAX_ITEMS) {
        const int nt = n_items - first < COPY_MAX_ITEMS ? n_items - first : COPY_MAX_ITEMS;
        int blocks = 0;
        for (int i = 0; i < nt; ++i) {
            const mkgnn_copy_item& s = items[first + i];
            if (!s.dst || !s.src || s.numel < 1 || s.numel > (1 << 30))
                return api_fail("mkgnn_flat_copy: item %d: null pointer or numel %lld out of range", first + i, (long long)s.numel);
            a.t[i] = CopyItem{s.dst, s.src, (int32_t)s.numel, 0};
            a.blk_start[i] = blocks;
            blocks += (int)((s.numel + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        a.blk_start[nt] = blocks;
        a.nt = nt;
        flat_copy_kernel<<<blocks, 256, 0, st>>>(a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_flat_copy", e);
}

extern "C" int64_t mkgnn_adamw_state_floats(int64_t numel) {
    return numel < 1 ? 0 : 2 * numel + 3 + (numel + ADAM_CHUNK - 1) / ADAM_CHUNK;
}

extern "C" int mkgnn_adamw_step(const mkgnn_adamw_tensor* tensors, int32_t n_tensors, const mkgnn_adamw_group* groups,
                                int32_t n_groups, void* stream) {
    if (n_tensors < 0 || n_groups < 1 || n_groups > ADAM_MAX_GROUPS)
        return api_fail("mkgnn_adamw_step: %d tensors, %d groups (1..%d groups)", n_tensors, n_groups, ADAM_MAX_GROUPS);
    if (n_tensors == 0) return 0;
    if (!tensors || !groups) return api_fail("mkgnn_adamw_step: null pointer");
    hipStream_t st = (hipStream_t)stream;
    AdamArgs a{};
    for (int g = 0; g < n_groups; ++g) {
        const mkgnn_adamw_group& s = groups[g];
        if (!(s.beta1 >= 0.f && s.beta1 < 1.f && s.beta2 >= 0.f && s.beta2 < 1.f) || s.eps < 0.f || s.weight_decay < 0.f)
            return api_fail("mkgnn_adamw_step: group %d has betas (%g, %g), eps %g, weight_decay %g", g, s.beta1, s.beta2, s.eps, s.weight_decay);
        a.grp[g] = AdamGroup{s.lr_device, s.lr, s.beta1, s.beta2, s.eps, s.weight_decay, s.maximize, s.grad_scale};
    }
    for (int32_t first = 0; first < n_tensors; first += ADAM_MAX_TENSORS) {
        const int nt = n_tensors - first < ADAM_MAX_TENSORS ? n_tensors - first : ADAM_MAX_TENSORS;
        int blocks = 0;
        for (int i = 0; i < nt; ++i) {
            const mkgnn_adamw_tensor& s = tensors[first + i];
            if (!s.param || !s.grad || !s.state || s.numel < 1 || s.numel > (1 << 30) || s.group < 0 || s.group >= n_groups)
                return api_fail("mkgnn_adamw_step: tensor %d: null pointer, numel %lld or group %d out of range", first + i, (long long)s.numel, s.group);
            a.t[i] = AdamTensor{s.param, s.grad, s.state, s.active, (int32_t)s.numel, s.group};
            a.blk_start[i] = blocks;
            blocks += (int)((s.numel + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        a.blk_start[nt] = blocks;
        a.nt = nt;
        adamw_step_kernel<<<blocks, 256, 0, st>>>(a);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_adamw_step", e);
}
