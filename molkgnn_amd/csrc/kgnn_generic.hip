// Generic (any F, E, L) gfx950 kernels of the molecular-kernel convolution.
//
// These are the shape-agnostic kernels: the bank preparation, the backward
// pass and a plain forward that serves every (F, E, L) the reference's
// constructors accept.  The MFMA forward for the shapes the model actually
// uses lives in kgnn_mfma.hip.  Wave = 64 lanes everywhere.
//
// Reference being computed: models/MolKGNN/kernels.py
//   calculate_total_score :353-425, permute :89-130, get_chirality_sign :279-350,
//   BaseKernelSetConv.forward :610-751.
#include "kgnn_launch.h"
#include "kgnn_prepare.h"

namespace mkgnn {

// ------------------------------------------------------------------ P1 ----
// 1 / max(||row||, eps); one wave per row.  Lane l owns columns 128 j + 2 l, + 1: the same
// assignment and summation order as segment_sum_rows_kernel's fused norm, so a norm handed over
// by the producer of x is bit-identical to the one computed here.
template <bool VEC2>
__global__ void row_inv_norm_kernel(const float* __restrict__ x, int64_t stride, int64_t n, int F,
                                    float* __restrict__ inv) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        float s = 0.f;
        for (int f = 2 * lane; f < F; f += 128) {
            float2 v;
            if constexpr (VEC2) v = *(const float2*)(x + r * stride + f);
            else { v.x = x[r * stride + f]; v.y = f + 1 < F ? x[r * stride + f + 1] : 0.f; }
            s = fmaf(v.x, v.x, s);
            if (f + 1 < F) s = fmaf(v.y, v.y, s);
        }
        s = wave_sum(s);
        if (lane == 0) inv[r] = 1.f / fmaxf(sqrtf(s), MKGNN_EPS);
    }
}

// ------------------------------------------------------------------ P2 ----

// Unit-normalise every kernel row, keep 1/norm, tabulate the support
// tetrahedron signs and the mixing weights.  One wave per task.
__global__ void bank_prepare_kernel(PrepArgs a) {
    const int task = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (task >= a.row_start[MKGNN_MAX_DEGREE]) return;
    bank_prepare_task(a, task);
}

// the banks of up to PREP_MANY_MAX forward calls (the layers of a model) in one launch: mkgnn_bank_prepare
//
// Round 6: blocks behind the preparation's own (blockIdx >= prep_blocks) READ the arrays of m.touch and throw the values away
// (mkgnn_touch_hint, taken here when no batch norm launch took it first: DESIGN 4.1g).
__global__ void bank_prepare_many_kernel(PrepManyArgs m) {
    if ((int)blockIdx.x >= m.prep_blocks) {
        touch_body(m.touch, blockIdx.x - m.prep_blocks, gridDim.x - m.prep_blocks);
        return;
    }
    bank_prepare_many_block(m, blockIdx.x);
}

// ------------------------------------------------------------- forward ----

// One wave per atom, lanes over kernels.  Plain and cache-served: this is the
// any-shape path, not the fast one.
template <int D>
__global__ void __launch_bounds__(256) kc_forward_generic(FwdArgs a) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float ws = a.mix[0], wc = a.mix[1], we = a.mix[2], wsum = a.mix[3];
    for (int64_t n = wave; n < a.n; n += nwaves) {
        const int64_t focal = a.sel[n];
        int64_t nb[D];
        float inb[D];
#pragma unroll
        for (int j = 0; j < D; ++j) { nb[j] = a.nei[n * D + j]; inb[j] = a.inv[nb[j]]; }
        const float ifoc = a.inv[focal];
        // neighbour bond vectors: 1 / max(|e|, eps)
        float ie[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float s = 0.f;
            for (int e = lane; e < a.E; e += 64) { float v = a.e_nei[(n * D + j) * a.E + e]; s = fmaf(v, v, s); }
            s = wave_sum(s);
            ie[j] = 1.f / fmaxf(sqrtf(s), MKGNN_EPS);
        }
        // chirality, atom side (kernels.py:305-337)
        bool not_chiral = false;
        float sign_nei = 0.f;
        const bool do_chir = (D == 4) && a.last;
        if constexpr (D == 4) {
            if (do_chir) {
                unsigned pair_diff = 0;   // bit k set: pair k differs somewhere in my lanes
                for (int f = lane; f < a.F; f += 64) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = a.x[nb[j] * a.xs + f];
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = i + 1; j < 4; ++j, ++k)
                            if (!(v[i] == v[j])) pair_diff |= 1u << k;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) pair_diff |= __shfl_xor((int)pair_diff, o, 64);
                not_chiral = (pair_diff != 0x3Fu);   // some pair is equal everywhere
                float t[3][3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        t[j][c] = __fsub_rn(a.p_nei[(n * 4 + j) * 3 + c], a.p_focal[n * 3 + c]);
                sign_nei = triple_sign(t[0], t[1], t[2]);
            }
        }
        for (int l = lane; l < a.L; l += 64) {
            float cm[D][D];
            float cc = 0.f;
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) cm[i][j] = 0.f;
            const float* cen = a.cen + (size_t)l * a.F;
            const float* sup = a.sup + (size_t)l * D * a.F;
            for (int f = 0; f < a.F; ++f) {
                float xf = a.x[focal * a.xs + f];
                cc = fmaf(xf, cen[f], cc);
                float sv[D];
#pragma unroll
                for (int j = 0; j < D; ++j) sv[j] = sup[j * a.F + f];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    float xv = a.x[nb[i] * a.xs + f];
#pragma unroll
                    for (int j = 0; j < D; ++j) cm[i][j] = fmaf(xv, sv[j], cm[i][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) cm[i][j] *= inb[i];
            cc *= ifoc;
            float best; int idx;
            best_permutation<D>(cm, best, idx);
            // edge score with the same permutation (kernels.py:382-390)
            float ed = 0.f;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const float* es = a.edg + ((size_t)l * D + perm_at<D>(idx, i)) * a.E;
                const float* en = a.e_nei + (n * D + i) * a.E;
                float dt = 0.f;
                for (int e = 0; e < a.E; ++e) dt = fmaf(en[e], es[e], dt);
                dt *= ie[i];
                ed = (i == 0) ? dt : __fadd_rn(ed, dt);
            }
            ed = div_by<D>(ed);
            float sc = __fadd_rn(__fadd_rn(__fmul_rn(best, ws), __fmul_rn(cc, wc)), __fmul_rn(ed, we)) / wsum;
            float ch = 1.f;
            if constexpr (D == 4) {
                if (do_chir && !not_chiral) ch = ((float)a.chir[l * 12 + idx] == sign_nei) ? 1.f : -1.f;
                sc *= ch;
            }
            a.out[focal * a.os + a.off + l] = sc;
            if (a.pair) pair_store(a.pair, (size_t)n * a.L + l, best, cc, ed, idx);
            if (a.chir_out) a.chir_out[(size_t)n * a.L + l] = (int8_t)ch;
        }
    }
}

// ------------------------------------------------------------ backward ----

// B1: gradient w.r.t. the unit rows of every (atom, slot): one wave per atom,
// lanes over features.  slot 0 = focal row, slot 1+a = neighbour a.
template <int D>
__global__ void __launch_bounds__(256) kc_backward_rows(BwdArgs a) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float ws = a.mix[0] / a.mix[3], wc = a.mix[1] / a.mix[3];
    for (int64_t n = wave; n < a.n; n += nwaves) {
        const int64_t focal = a.sel[n];
        for (int f0 = 0; f0 < a.F; f0 += 64) {
            const int f = f0 + lane;
            float acc[D + 1];
#pragma unroll
            for (int s = 0; s <= D; ++s) acc[s] = 0.f;
            for (int l = 0; l < a.L; ++l) {
                float g = a.gout[focal * a.gs + a.off + l];
                if (a.chir) g *= (float)a.chir[(size_t)n * a.L + l];
                const int idx = pair_index(a.pair, (size_t)n * a.L + l);
                if (f < a.F) {
                    acc[0] = fmaf(g * wc, a.cen[(size_t)l * a.F + f], acc[0]);
                    const float gsup = g * ws / (float)D;
#pragma unroll
                    for (int s = 0; s < D; ++s)
                        acc[1 + s] = fmaf(gsup, a.sup[((size_t)l * D + perm_at<D>(idx, s)) * a.F + f], acc[1 + s]);
                }
            }
            if (f < a.F) {
#pragma unroll
                for (int s = 0; s <= D; ++s)
                    a.contrib[(a.contrib_base + n * (D + 1) + s) * a.CS + f] = acc[s];
            }
        }
    }
}

// B2: partial gradients of the unit kernel rows.  grid = (bank rows, chunks of
// atoms); lanes over features.  Row order of a bank gradient:
//   [0, L)            centre rows            (width F)
//   [L, L + L*D)      support rows (l, b)    (width F)
//   [L+L*D, L+2*L*D)  edge rows (l, b)       (width E)
//   then 3 score-weight partials.
template <int D>
__global__ void __launch_bounds__(128) kc_backward_bank(BwdArgs a) {
    const int r = blockIdx.x;
    const int c = blockIdx.y;
    const int L = a.L;
    const int64_t lo = a.n * c / a.nchunk, hi = a.n * (c + 1) / a.nchunk;
    const size_t bank_fl = bank_floats(D, L, a.F, a.E);
    float* slab = a.slab + (size_t)c * bank_fl;
    const float ws = a.mix[0] / a.mix[3], wc = a.mix[1] / a.mix[3], we = a.mix[2] / a.mix[3];
    const int nrow = L + 2 * L * D;
    if (r == nrow) {
        // score-weight partials: d sc / d theta_k = w_k (score_k - sc) / W  
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        for (int64_t n = lo; n < hi; ++n) {
            const int64_t focal = a.sel[n];
            for (int l = threadIdx.x; l < L; l += blockDim.x) {
                float g = a.gout[focal * a.gs + a.off + l];
                if (a.chir) g *= (float)a.chir[(size_t)n * a.L + l];
                const mkgnn_f32x4 rec = pair_load(a.pair, (size_t)n * L + l);
                const float S = rec[0], C = rec[1], Ed = rec[2];
                float sc = (S * a.mix[0] + C * a.mix[1] + Ed * a.mix[2]) / a.mix[3];
                p0 = fmaf(g * ws, S - sc, p0);
                p1 = fmaf(g * wc, C - sc, p1);
                p2 = fmaf(g * we, Ed - sc, p2);
            }
        }
        __shared__ float red[3][2];
        p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[0][w] = p0; red[1][w] = p1; red[2][w] = p2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            size_t o = (size_t)L * a.F + (size_t)L * D * a.F + (size_t)L * D * a.E;
            slab[o + 0] = red[0][0] + red[0][1];
            slab[o + 1] = red[1][0] + red[1][1];
            slab[o + 2] = red[2][0] + red[2][1];
        }
        return;
    }
    int l, b, kind;   // kind 0 centre, 1 support, 2 edge
    size_t out_off;
    if (r < L) { kind = 0; l = r; b = 0; out_off = (size_t)r * a.F; }
    else if (r < L + L * D) { kind = 1; l = (r - L) / D; b = (r - L) % D; out_off = (size_t)L * a.F + (size_t)(r - L) * a.F; }
    else { kind = 2; l = (r - L - L * D) / D; b = (r - L - L * D) % D;
           out_off = (size_t)L * a.F + (size_t)L * D * a.F + (size_t)(r - L - L * D) * a.E; }
    const int width = (kind == 2) ? a.E : a.F;
    for (int f = threadIdx.x; f < width; f += blockDim.x) {
        float acc = 0.f;
        for (int64_t n = lo; n < hi; ++n) {
            const int64_t focal = a.sel[n];
            float g = a.gout[focal * a.gs + a.off + l];
            if (a.chir) g *= (float)a.chir[(size_t)n * a.L + l];
            if (kind == 0) {
                acc = fmaf(g * wc * a.inv[focal], a.x[focal * a.xs + f], acc);
            } else {
                const int idx = pair_index(a.pair, (size_t)n * a.L + l);
                // which neighbour slot was matched to support b: pi(slot) == b
                int slot = 0;
#pragma unroll
                for (int s = 0; s < D; ++s) if (perm_at<D>(idx, s) == b) slot = s;
                if (kind == 1) {
                    const int64_t nbr = a.nei[n * D + slot];
                    acc = fmaf(g * ws / (float)D * a.inv[nbr], a.x[nbr * a.xs + f], acc);
                } else {
                    const float* en = a.e_nei + (n * D + slot) * a.E;
                    float s2 = 0.f;
                    for (int e = 0; e < a.E; ++e) s2 = fmaf(en[e], en[e], s2);
                    float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
                    acc = fmaf(g * we / (float)D * ie, en[f], acc);
                }
            }
        }
        slab[out_off + f] = acc;
    }
}


// Sum the per-block partial slabs in a fixed order and undo the unit normalisation:
// d/ds of s/max(|s|,eps):  (g - (g . s^) s^) / |s|   (or g / eps below eps).
// One block per bank row, all four degrees in one launch (four separate launches of this latency-bound
// kernel cost ~9 us each).  Wave w of 8 sums chunks w, w + 8, ... with 16 loads in flight; the partial sums
// are combined in an order fixed by the code -- the same on every run.
__global__ void __launch_bounds__(512) kc_backward_bank_reduce(BankReduceAllArgs all) {
    int di = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if ((int)blockIdx.x >= all.blk_start[k]) di = k;
    const BankReduceArgs& a = all.deg[di];
    const int D = di + 1;
    const int r = blockIdx.x - all.blk_start[di];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = 8;
    const int L = a.L;
    const size_t bank_fl = bank_floats(D, L, a.F, a.E);
    const int nrow = L + 2 * L * D;
    __shared__ float part[NWV][256];
    if (r == nrow) {
        // score-weight partials: thread t sums entries t, t+512, ...; then a fixed-order tree over the block
        float* red = &part[0][0];
        for (int k = 0; k < 3; ++k) {
            float s = 0.f;
            for (int c = tid; c < a.theta_count; c += 512) s += a.theta_src[(size_t)c * a.theta_stride + k];
            red[tid] = s;
            __syncthreads();
            for (int w = 256; w > 0; w >>= 1) {
                if (tid < w) red[tid] += red[tid + w];
                __syncthreads();
            }
            if (tid == 0) {
                float* dst = k == 0 ? a.g.support_attr_sc_weight : (k == 1 ? a.g.center_attr_sc_weight : a.g.edge_attr_support_sc_weight);
                if (dst) *dst = red[0];
            }
            __syncthreads();
        }
        return;
    }
    const float* unit; float inv; float* dst; size_t off; int width;
    if (r < L) { unit = a.cen + (size_t)r * a.F; inv = a.icen[r]; dst = a.g.x_center ? a.g.x_center + (size_t)r * a.F : nullptr; off = (size_t)r * a.F; width = a.F; }
    else if (r < L + L * D) { int q = r - L; unit = a.sup + (size_t)q * a.F; inv = a.isup[q]; dst = a.g.x_support ? a.g.x_support + (size_t)q * a.F : nullptr; off = (size_t)L * a.F + (size_t)q * a.F; width = a.F; }
    else { int q = r - L - L * D; unit = a.edg + (size_t)q * a.E; inv = a.iedg[q]; dst = a.g.edge_attr_support ? a.g.edge_attr_support + (size_t)q * a.E : nullptr; off = (size_t)L * a.F + (size_t)L * D * a.F + (size_t)q * a.E; width = a.E; }
    if (!dst) return;
    for (int f = lane; f < width; f += 64) {
        float s[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u] = 0.f;
        const float* base = a.slab + off + f;
        for (int c = wave; c < a.nchunk; c += NWV * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {       // unconditional clamped loads, masked sums
                const int cc = c + NWV * u;
                v[u] = base[(size_t)(cc < a.nchunk ? cc : c) * bank_fl];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (c + NWV * u < a.nchunk) s[u] += v[u];
        }
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) s[u] += s[u + w];
        part[wave][f] = s[0];
    }
    __syncthreads();
    if (wave != 0) return;
    float dotp = 0.f;
    for (int f = lane; f < width; f += 64) {
        const float s = ((part[0][f] + part[1][f]) + (part[2][f] + part[3][f])) + ((part[4][f] + part[5][f]) + (part[6][f] + part[7][f]));
        part[0][f] = s;
        dotp = fmaf(s, unit[f], dotp);
    }
    dotp = wave_sum(dotp);
    const bool clamped = inv >= (1.f / MKGNN_EPS);
    for (int f = lane; f < width; f += 64) {
        const float s = part[0][f];
        dst[f] = clamped ? s * inv : (s - dotp * unit[f]) * inv;
    }
}

// B3: per atom, sum the contribution rows that point at it (fixed CSR order -> reproducible) and
// undo the row normalisation.  One wave per atom, two floats per lane; the row ids of the segment
// are fetched first and all row loads issued together, so one atom costs ~2 memory round trips.
__global__ void __launch_bounds__(256) kc_backward_gather(const float* __restrict__ contrib, int64_t cs, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ rows, const float* __restrict__ x,
                                                          int64_t xs, const float* __restrict__ inv, int64_t n, int F,
                                                          float* __restrict__ gx, int64_t gxs) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int SEG = 8;                 // rows fetched per round (focal + up to 4 neighbour roles fit in one)
    for (int64_t j = wave; j < n; j += nwaves) {
        const int lo = rowptr[j], hi = rowptr[j + 1];
        const float iv = inv[j];
        const bool clamped = iv >= (1.f / MKGNN_EPS);
        for (int f0 = 0; f0 < F; f0 += 128) {
            const int f = f0 + 2 * lane;
            const bool two = f + 1 < F, one = f < F;
            float2 acc = {0.f, 0.f};
            for (int k0 = lo; k0 < hi; k0 += SEG) {
                int rid[SEG];
#pragma unroll
                for (int u = 0; u < SEG; ++u) rid[u] = k0 + u < hi ? rows[k0 + u] : -1;
                float2 v[SEG];
#pragma unroll
                for (int u = 0; u < SEG; ++u) {
                    v[u] = float2{0.f, 0.f};
                    if (rid[u] >= 0 && one) {
                        const float* src = contrib + (size_t)rid[u] * cs + f;
                        if (two && ((F & 1) == 0)) v[u] = *(const float2*)src;
                        else { v[u].x = src[0]; if (two) v[u].y = src[1]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < SEG; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
            }
            float2 xv = {0.f, 0.f};
            if (one) { xv.x = x[j * xs + f] * iv; if (two) xv.y = x[j * xs + f + 1] * iv; }
            // the dot product spans the whole row; F <= 256 means at most two passes, handled by accumulating
            float dotp = acc.x * xv.x + acc.y * xv.y;
            if (F > 128) {
                // second half of a wide row contributes to the same dot product: fetch it now
                const int f2 = (f0 == 0 ? 128 : 0) + 2 * lane;
                float2 a2 = {0.f, 0.f}, x2 = {0.f, 0.f};
                if (f2 < F) {
                    for (int k = lo; k < hi; ++k) {
                        const float* src = contrib + (size_t)rows[k] * cs + f2;
                        a2.x += src[0];
                        if (f2 + 1 < F) a2.y += src[1];
                    }
                    x2.x = x[j * xs + f2] * iv;
                    if (f2 + 1 < F) x2.y = x[j * xs + f2 + 1] * iv;
                }
                dotp += a2.x * x2.x + a2.y * x2.y;
            }
            dotp = wave_sum(dotp);
            if (one) {
                gx[j * gxs + f] = clamped ? acc.x * iv : (acc.x - dotp * xv.x) * iv;
                if (two) gx[j * gxs + f + 1] = clamped ? acc.y * iv : (acc.y - dotp * xv.y) * iv;
            }
        }
    }
}

// ---------------------------------------------------------- propagate ----
// out[i, :] = sum_k in[col[k], :] over the CSR segment of row i (KernelLayer.py:119-123 with
// aggr='add'); optionally also 1 / max(|out[i]|, eps), which the next layer's cosine needs anyway.
// One wave per row, two floats per lane; the segment's column ids are fetched first and all row
// loads issued together.  Sums run in CSR order: reproducible.
template <bool VEC2>
__global__ void __launch_bounds__(256) segment_sum_rows_kernel(const float* __restrict__ in, int64_t is,
                                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               int64_t n, int width, float* __restrict__ out, int64_t os,
                                                               float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int SEG = 8;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int lo = rowptr[i], hi = rowptr[i + 1];
        float ss = 0.f;
        for (int f0 = 0; f0 < width; f0 += 128) {
            const int f = f0 + 2 * lane;
            const bool one = f < width, two = f + 1 < width;
            const int fc = one ? f : 0;                      // clamped: loads stay unconditional
            float2 acc = {0.f, 0.f};
            for (int k0 = lo; k0 < hi; k0 += SEG) {
                int cid[SEG];
#pragma unroll
                for (int u = 0; u < SEG; ++u) cid[u] = col[k0 + u < hi ? k0 + u : hi - 1];
                float2 v[SEG];
#pragma unroll
                for (int u = 0; u < SEG; ++u) {
                    const float* src = in + (size_t)cid[u] * is + fc;
                    if constexpr (VEC2) v[u] = *(const float2*)src;
                    else { v[u].x = src[0]; v[u].y = two ? src[1] : 0.f; }
                }
#pragma unroll
                for (int u = 0; u < SEG; ++u)
                    if (k0 + u < hi) { acc.x += v[u].x; acc.y += v[u].y; }
            }
            if (one) {
                if constexpr (VEC2) { if (two) *(float2*)(out + i * os + f) = acc; else out[i * os + f] = acc.x; }
                else { out[i * os + f] = acc.x; if (two) out[i * os + f + 1] = acc.y; }
                ss = fmaf(acc.x, acc.x, ss);
                if (two) ss = fmaf(acc.y, acc.y, ss);
            }
        }
        if (inv_norm) {
            ss = wave_sum(ss);
            if (lane == 0) inv_norm[i] = 1.f / fmaxf(sqrtf(ss), MKGNN_EPS);
        }
    }
}

// ------------------------------------------------------------ launchers ---
static inline int grid_for_waves(int64_t waves, int threads = 256) {
    int64_t per_block = threads / 64;
    int64_t blocks = (waves + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (int)blocks;
}

hipError_t launch_row_inv_norm(const float* x, int64_t stride, int64_t n, int F, float* inv, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (try_row_inv_norm_aligned(x, stride, n, F, inv, st, &e)) return e;
    if (stride % 2 == 0 && F % 2 == 0 && (uintptr_t)x % 8 == 0)
        row_inv_norm_kernel<true><<<grid_for_waves(n), 256, 0, st>>>(x, stride, n, F, inv);
    else
        row_inv_norm_kernel<false><<<grid_for_waves(n), 256, 0, st>>>(x, stride, n, F, inv);
    return hipGetLastError();
}

static int fill_prep_args(PrepArgs& a, const mkgnn_kernel_bank banks[4], const WorkspaceLayout& w, char* ws, int F, int E);

void build_bank_prepare_many(int count, const mkgnn_kernel_bank* banks, const WorkspaceLayout* w, char* const* ws, const int* F,
                             int E, PrepManyArgs* out) {
    PrepManyArgs& m = *out;
    m.count = count;
    m.task_start[0] = 0;
    for (int k = 0; k < count; ++k) m.task_start[k + 1] = m.task_start[k] + fill_prep_args(m.layer[k], banks + 4 * k, w[k], ws[k], F[k], E);
    const int tasks = m.task_start[count];
    m.prep_blocks = (tasks + 3) / 4;
    m.touch.count = 0;
}

hipError_t launch_bank_prepare_args(const PrepManyArgs& m, hipStream_t st) {
    const int blocks = m.prep_blocks + (m.touch.count > 0 ? TOUCH_BLOCKS : 0);
    if (blocks == 0) return hipSuccess;
    bank_prepare_many_kernel<<<blocks, 256, 0, st>>>(m);
    return hipGetLastError();
}

hipError_t launch_bank_prepare_many(int count, const mkgnn_kernel_bank* banks, const WorkspaceLayout* w, char* const* ws,
                                    const int* F, int E, hipStream_t st, const TouchArgs* touch) {
    PrepManyArgs m;
    build_bank_prepare_many(count, banks, w, ws, F, E, &m);
    if (touch) m.touch = *touch;
    return launch_bank_prepare_args(m, st);
}

hipError_t launch_bank_prepare(const mkgnn_kernel_bank banks[4], const WorkspaceLayout& w, char* ws, int F, int E,
                               hipStream_t st) {
    PrepArgs a;
    const int tasks = fill_prep_args(a, banks, w, ws, F, E);
    if (tasks == 0) return hipSuccess;
    bank_prepare_kernel<<<(tasks + 3) / 4, 256, 0, st>>>(a);
    return hipGetLastError();
}

static int fill_prep_args(PrepArgs& a, const mkgnn_kernel_bank banks[4], const WorkspaceLayout& w, char* ws, int F, int E) {
    a.F = F; a.E = E;
    a.row_start[0] = 0;
    for (int i = 0; i < 4; ++i) {
        a.bank[i] = banks[i];
        a.cen[i] = (float*)(ws + w.bank[i].cen);   a.sup[i] = (float*)(ws + w.bank[i].sup);
        a.edg[i] = (float*)(ws + w.bank[i].edg);   a.icen[i] = (float*)(ws + w.bank[i].icen);
        a.isup[i] = (float*)(ws + w.bank[i].isup); a.iedg[i] = (float*)(ws + w.bank[i].iedg);
        a.chir[i] = (int8_t*)(ws + w.bank[i].chir); a.mix[i] = (float*)(ws + w.bank[i].mix);
        a.padded[i] = (float*)(ws + w.bank[i].padded); a.edge_padded[i] = (float*)(ws + w.bank[i].edge_padded);
        int L = banks[i].num_kernels, d = i + 1;
        a.row_start[i + 1] = a.row_start[i] + (L > 0 ? L + 2 * L * d + 1 + (d == 4 ? (12 * L + 63) / 64 : 0) : 0);
    }
    return a.row_start[4];
}

template <int D>
static hipError_t launch_fwd_d(const FwdArgs& a, hipStream_t st) {
    kc_forward_generic<D><<<grid_for_waves(a.n), 256, 0, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_forward_generic(int d, const FwdArgs& a, hipStream_t st) {
    if (a.n == 0 || a.L == 0) return hipSuccess;
    switch (d) {
        case 1: return launch_fwd_d<1>(a, st);
        case 2: return launch_fwd_d<2>(a, st);
        case 3: return launch_fwd_d<3>(a, st);
        default: return launch_fwd_d<4>(a, st);
    }
}

template <int D>
static hipError_t launch_bwd_d(const BwdArgs& a, hipStream_t st) {
    if (a.n > 0) {
        kc_backward_rows<D><<<grid_for_waves(a.n), 256, 0, st>>>(a);
        dim3 grid(a.L + 2 * a.L * D + 1, a.nchunk);
        kc_backward_bank<D><<<grid, 128, 0, st>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_backward_generic(int d, const BwdArgs& a, hipStream_t st) {
    if (a.L == 0) return hipSuccess;
    switch (d) {
        case 1: return launch_bwd_d<1>(a, st);
        case 2: return launch_bwd_d<2>(a, st);
        case 3: return launch_bwd_d<3>(a, st);
        default: return launch_bwd_d<4>(a, st);
    }
}

hipError_t launch_bank_reduce_all(const BankReduceArgs r[4], hipStream_t st) {
    BankReduceAllArgs all;
    int blk = 0;
    for (int i = 0; i < 4; ++i) {
        all.deg[i] = r[i];
        all.blk_start[i] = blk;
        if (r[i].L > 0) blk += r[i].L + 2 * r[i].L * (i + 1) + 1;
    }
    if (blk == 0) return hipSuccess;
    kc_backward_bank_reduce<<<blk, 512, 0, st>>>(all);
    return hipGetLastError();
}

hipError_t launch_backward_gather(const float* contrib, int64_t cs, int64_t n_contrib_rows, const int32_t* rowptr,
                                  const int32_t* rows, const float* x, int64_t xs, const float* inv, int64_t n, int F,
                                  float* gx, int64_t gxs, bool allow_fast, hipStream_t st, bool x_split) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (allow_fast && n_contrib_rows > 0 && try_backward_gather_aligned(contrib, cs, rowptr, rows, x, xs, inv, n, F, gx, gxs, st, &e, x_split)) return e;
    if (x_split) return hipErrorInvalidValue;              // (pre-split rows: only the pipelined gather reads them; the caller checks the shapes first)
    kc_backward_gather<<<grid_for_waves(n), 256, 0, st>>>(contrib, cs, rowptr, rows, x, xs, inv, n, F, gx, gxs);
    return hipGetLastError();
}

hipError_t launch_segment_sum(const float* in, int64_t is, const int32_t* rowptr, const int32_t* col, int64_t n,
                              int width, float* out, int64_t os, float* inv_norm, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (try_segment_sum_aligned(in, is, rowptr, col, n, width, out, os, inv_norm, st, &e)) return e;
    // 8-byte accesses need even strides / width and 8-byte aligned bases
    const bool vec2 = (is % 2 == 0) && (os % 2 == 0) && (width % 2 == 0) && (((uintptr_t)in | (uintptr_t)out) % 8 == 0);
    if (vec2) segment_sum_rows_kernel<true><<<grid_for_waves(n), 256, 0, st>>>(in, is, rowptr, col, n, width, out, os, nullptr);
    else segment_sum_rows_kernel<false><<<grid_for_waves(n), 256, 0, st>>>(in, is, rowptr, col, n, width, out, os, nullptr);
    e = hipGetLastError();
    // the norm handed to the next layer must be the one mkgnn_row_inv_norm would compute on `out`
    // (same kernel choice, same summation order), whatever layout `in` had
    if (e == hipSuccess && inv_norm) e = launch_row_inv_norm(out, os, n, width, inv_norm, st);
    return e;
}

}  // namespace mkgnn
