// Molecule-resident small-batch step (include/molkgnn_hip.h, "Molecule-resident small-batch step"; ABI v4).
//
// One workgroup owns a chunk of whole molecules (<= 64 atoms): its atom rows live in LDS from the batch norm to the graph
// embedding and back.  Per layer the d x d cosine matrices of every (atom, kernel) pair come out of ONE dense product
//
//      St [atoms x bank rows] = X [atoms x F] . BankU^T [F x bank rows]            (fp32 MFMA 16x16x4, exact fma chains)
//
// of the chunk's rows with ALL unit-normalised kernel rows of the layer (supports of every degree and centres), followed by a
// per-pair gather  cm[a][b] = St[nei(n, a)][support (d, l, b)]  -- inside a molecule a neighbour's row IS one of the chunk's
// rows, so nothing is gathered before the product.  The backward runs the adjoint: the pairs' coefficients are scattered
// (deterministically: every entry has one owner thread) into Cf [atoms x bank rows], and two dense products give
//      d loss / d unit atom rows   G  = Cf  . BankU            [atoms x F]
//      d loss / d unit bank rows   Bg = Cf^T . U               [bank rows x F]   (per-workgroup slab, summed afterwards)
// Reference: models/MolKGNN/kernels.py:353-425 (scores), :279-350 (chirality), :610-751 (KernelSetConv),
// KernelLayer.py:109-123 (layer loop, propagate), MolKGNNNet.py:115-146 (batch norm, readout), model.py:150,169,190-198 (head).
#include "kgnn_launch.h"

namespace mkgnn {

typedef mkgnn_f32x4 v4;

constexpr int MOL_THREADS = 512;
constexpr int MOL_NW = MOL_THREADS / 64;   // waves of a workgroup
constexpr int MOL_TPW = (16 + MOL_NW - 1) / MOL_NW;   // bank-row tiles (<= 16 per pass) a wave takes in the forward products
constexpr int MOL_KSPLIT = MOL_NW / 8;     // the rows gradient's contraction is split over this many waves per feature tile
constexpr int MOL_GPW = 16 / MOL_KSPLIT;   // 16-row groups of a pass per wave there
constexpr int MOL_RS = 260;            // row stride of St / Cf (floats): <= 256 bank rows per pass; 260 = 4 * 65 keeps the 16
                                       // rows of a ds_read_b128 on distinct banks
constexpr int MOL_XS = 116;            // row stride of the atom-row buffers (112 + 4)
constexpr int MOL_MAXA = MKGNN_MOLECULE_MAX_ATOMS;
constexpr int MOL_MAXM = MKGNN_MOLECULE_MAX_MOLS;
constexpr int MOL_BN_BLOCKS = 32;      // partial-statistics blocks of the preparation launch
constexpr int MOL_HS = 68;             // row stride of the readout's pre-activation rows (H <= 64)

struct MolLayer {
    int F, FP, KJ;                     // input width, padded width (32 / 112), FP / 16
    int L[4], off[4], K;               // kernels per degree, their column offsets, their sum
    int sup_row[4], cen_row[4];        // row (inside its pass) of support (b = 0, l = 0) / centre (l = 0) of degree d
    int pass_rows[2], row_base[2];     // rows of a pass (multiple of 16) and its first row in bankU; pass 0: d = 1..3, pass 1: d = 4
    int rows_real[2];
    int RPT;                           // pass_rows[0] + pass_rows[1]
    int e_row[4], ER;                  // edge-support rows (order d, b, l)
    const float* bankU;                // [RPT, FP]  unit rows, zero padded
    const float* bankUT;               // [FP, RPT]
    const float* edgeU;                // [ER, 8]
    const int8_t* chir;                // [L_4, 12] sign of the support tetrahedron per order
    const float* mix;                  // [4][4] w_support, w_center, w_edge, their sum
    float* pair[4];                    // pair records (mkgnn_saved)
    int8_t* chir_out;                  // [N_4, L_4] or null
    float* x_save;                     // [n_atoms, FP] this layer's input rows (kept for the backward half of the launch)
    float* sim_out; int64_t sim_stride;
    size_t slab_bank, slab_edge;       // offsets (floats) inside a chunk's slab
    int theta_slot;                    // offset (floats) inside the chunk's small slab: [4][3]
};

struct MolArgs {
    int64_t n_atoms; int n_mols, n_chunks;
    const int32_t* chunk_ptr; const int64_t* mol_ptr; const int8_t* atom_deg; const int32_t* atom_rank;
    const int64_t* nei[4]; const float* eunit[4]; const float* p_focal4; const float* nei_p4;
    const float* x; int64_t xs; int F0;
    // batch norm
    const float *bn_w, *bn_b; float *run_mean, *run_var; int64_t* nbt; float bn_eps, bn_mom; int bn_training;
    const float* bn_part; int bn_nblk;
    int nl; MolLayer layer[MKGNN_MOLECULE_MAX_LAYERS];
    // readout
    const float *w1p, *w1pt; const float *b1, *w2, *b2; int H, G, HP, K3;
    // head
    int mode; const float *ffn_w, *ffn_b, *y; float head_drop; const int64_t* rng;
    const float* demb; float* emb; float* pred;
    // partial outputs
    // small read-only tables copied into LDS at the start of a chunk when they fit behind the buffers (the NT = 2 variant: unit
    // edge-support rows of every layer, lin1 padded and transposed, lin2): a global load behind every use otherwise
    int cache_on, c_edge[MKGNN_MOLECULE_MAX_LAYERS], c_w1p, c_w1pt, c_w2, cache_floats;
    unsigned long long* stamps;        // diagnostics (mkgnn_debug_molecule_stamps): cycle stamps of chunk 0's phases
    float* slab; size_t slab_floats;   // per chunk: [small | dW1 | per layer: bank, edge]
    int s_loss, s_ffn, s_lin2, s_lin1b, s_bn; size_t s_dw1;      // offsets inside the small slab / the slab
};

struct PrepLayer {
    mkgnn_kernel_bank bank[4];
    int F, FP, RPT, ER, E;
    int L[4];
    int sup_row[4], cen_row[4], pass_rows[2], row_base[2], e_row[4];
    float *bankU, *bankUT, *edgeU, *mix; int8_t* chir;
    int task0;                         // first task of the layer; order: RPT bank rows, ER edge rows, 1 mix, chirality tasks
};
struct PrepArgsMol {
    int nl; PrepLayer layer[MKGNN_MOLECULE_MAX_LAYERS];
    int task_w1, task_end;             // lin1 rows (HP tasks)
    const float* w1; int H, HP, K3; float *w1p, *w1pt;
    // batch-norm partial statistics (blocks behind the task blocks)
    int task_blocks; const float* x; int64_t xs; int64_t n; int C; float* bn_part; int bn_nblk;
    // edge_batch_norm's statistics (mkgnn_molecule_net.edge_stats): ONE more block, es_block (-1: none)
    int es_block; mkgnn_bn_stats es;
};
constexpr int64_t MOL_EDGE_STATS_ROWS = 8192;     // bond rows one block takes in the preparation launch (above: the caller's own launch)

// -------------------------------------------------------------------------------------------- preparation ----
__device__ __forceinline__ void prep_unit_row(const float* src, int width, int lane, float& m0, float& m1, float& iv) {
    const float v0 = src ? src[lane < width ? lane : 0] : 0.f, v1 = src ? src[lane + 64 < width ? lane + 64 : 0] : 0.f;
    m0 = (src && lane < width) ? v0 : 0.f; m1 = (src && lane + 64 < width) ? v1 : 0.f;
    float s = fmaf(m0, m0, 0.f);
    s = fmaf(m1, m1, s);
    s = wave_sum(s);
    iv = 1.f / fmaxf(sqrtf(s), MKGNN_EPS);
}

// column totals over the block's 32 row lanes (thread t: column t & 7, row lane t >> 3) -> res[0..7], in a fixed order
__device__ __forceinline__ void prep_col8_total(float v, float* sh, float* res) {
    const int t = threadIdx.x;
    __syncthreads();
    sh[t] = v;
    __syncthreads();
    if (t < 8) {
        float tot = 0.f;
        for (int k = 0; k < 32; ++k) tot += sh[k * 8 + t];
        res[t] = tot;
    }
    __syncthreads();
}

// What BatchNorm1d's training-mode call does to the module's buffers, for the bond rows (reference MolKGNNNet.py:116; the
// normalised rows themselves are never read): mean and centred squares in two passes over rows that stay in
// cache, then running_mean / running_var / num_batches_tracked as mkgnn_batchnorm_update_stats moves them.
__device__ __forceinline__ void prep_edge_stats(const mkgnn_bn_stats& e) {
    __shared__ float sh[256], tot[8], sq[8], cn[8];
    const int t = threadIdx.x, c = t & 7, rs = t >> 3;
    const int cc = c < e.C ? c : 0;
    const int64_t lim = e.row_key ? e.key_limit[0] : 0;
    float s = 0.f, n = 0.f;
    for (int64_t r = rs; r < e.n_rows; r += 32)
        if (!e.row_key || e.row_key[r] < lim) { s += e.x[r * e.x_stride + cc]; n += 1.f; }
    prep_col8_total(s, sh, tot);
    prep_col8_total(n, sh, cn);
    const float cnt = cn[0];
    const float mu = cnt > 0.f ? tot[c] / cnt : 0.f;
    float m2 = 0.f;
    for (int64_t r = rs; r < e.n_rows; r += 32)
        if (!e.row_key || e.row_key[r] < lim) { const float d = e.x[r * e.x_stride + cc] - mu; m2 = fmaf(d, d, m2); }
    prep_col8_total(m2, sh, sq);
    if (t < e.C && cnt > 0.f) {
        const float var = sq[t] / cnt;
        if (e.running_mean) e.running_mean[t] = fmaf(e.momentum, mu - e.running_mean[t], e.running_mean[t]);
        if (e.running_var) {
            const float unbiased = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
            e.running_var[t] = fmaf(e.momentum, unbiased - e.running_var[t], e.running_var[t]);
        }
    }
    if (t == 0 && e.num_batches_tracked) e.num_batches_tracked[0] += 1;
}

__global__ void __launch_bounds__(256) molecule_prepare_kernel(PrepArgsMol a) {
    if ((int)blockIdx.x == a.es_block) { prep_edge_stats(a.es); return; }
    if ((int)blockIdx.x >= a.task_blocks) {
        // ---- partial batch-norm statistics of block b: column sums and centred squares of its rows (two passes over
        // rows that stay in cache); combined in a fixed order by every workgroup of the step kernel
        __shared__ float sh[256], mean[32];
        const int b = (int)blockIdx.x - a.task_blocks;
        const int t = threadIdx.x, c = t & 31, rs = t >> 5;
        const int64_t per = (a.n + a.bn_nblk - 1) / a.bn_nblk;
        const int64_t lo = per * b, hi = lo + per < a.n ? lo + per : a.n;
        const int cc = c < a.C ? c : 0;
        float s = 0.f;
        for (int64_t r = lo + rs; r < hi; r += 8) s += a.x[r * a.xs + cc];
        sh[t] = s;
        __syncthreads();
        if (rs == 0) {
            float tot = 0.f;
            for (int k = 0; k < 8; ++k) tot += sh[k * 32 + c];
            mean[c] = hi > lo ? tot / (float)(hi - lo) : 0.f;
            if (c < a.C) a.bn_part[(size_t)b * 2 * a.C + c] = tot;
        }
        __syncthreads();
        const float mu = mean[c];
        float m2 = 0.f;
        for (int64_t r = lo + rs; r < hi; r += 8) { const float d = a.x[r * a.xs + cc] - mu; m2 = fmaf(d, d, m2); }
        __syncthreads();
        sh[t] = m2;
        __syncthreads();
        if (rs == 0 && c < a.C) {
            float tot = 0.f;
            for (int k = 0; k < 8; ++k) tot += sh[k * 32 + c];
            a.bn_part[(size_t)b * 2 * a.C + a.C + c] = tot;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int task = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (task >= a.task_end) return;
    if (task >= a.task_w1) {
        // lin1 rows, zero padded to [HP, 112], and their transpose [112, HP]
        const int h = task - a.task_w1;
        const float* src = h < a.H ? a.w1 + (size_t)h * a.K3 : nullptr;
        const float v0 = (src && lane < a.K3) ? src[lane] : 0.f, v1 = (src && lane + 64 < a.K3) ? src[lane + 64] : 0.f;
        a.w1p[(size_t)h * 112 + lane] = v0;
        if (lane + 64 < 112) a.w1p[(size_t)h * 112 + lane + 64] = v1;
        a.w1pt[(size_t)lane * a.HP + h] = v0;
        if (lane + 64 < 112) a.w1pt[(size_t)(lane + 64) * a.HP + h] = v1;
        return;
    }
    int li = 0;
    while (li + 1 < a.nl && task >= a.layer[li + 1].task0) ++li;
    const PrepLayer& P = a.layer[li];
    int r = task - P.task0;
    if (r < P.RPT) {
        // bank row r: which (degree, support slot / centre, kernel), or padding
        const int p = r >= P.row_base[1] ? 1 : 0;
        const int rr = r - P.row_base[p];
        const float* src = nullptr;
        for (int i = (p ? 3 : 0); i < (p ? 4 : 3); ++i) {
            const int d = i + 1, L = P.L[i];
            if (rr >= P.sup_row[i] && rr < P.sup_row[i] + d * L) {
                const int b = (rr - P.sup_row[i]) / L, l = (rr - P.sup_row[i]) - b * L;
                src = P.bank[i].x_support + ((size_t)l * d + b) * P.F;
            } else if (rr >= P.cen_row[i] && rr < P.cen_row[i] + L) {
                src = P.bank[i].x_center + (size_t)(rr - P.cen_row[i]) * P.F;
            }
        }
        float m0, m1, iv;
        prep_unit_row(src, P.F, lane, m0, m1, iv);
        if (lane < P.FP) { P.bankU[(size_t)r * P.FP + lane] = m0 * iv; P.bankUT[(size_t)lane * P.RPT + r] = m0 * iv; }
        if (lane + 64 < P.FP) { P.bankU[(size_t)r * P.FP + lane + 64] = m1 * iv; P.bankUT[(size_t)(lane + 64) * P.RPT + r] = m1 * iv; }
        return;
    }
    r -= P.RPT;
    if (r < P.ER) {
        int i = 3;
        while (i > 0 && r < P.e_row[i]) --i;
        const int d = i + 1, L = P.L[i];
        const int b = (r - P.e_row[i]) / L, l = (r - P.e_row[i]) - b * L;
        const float* src = P.bank[i].edge_attr_support + ((size_t)l * d + b) * P.E;
        float m0, m1, iv;
        prep_unit_row(src, P.E, lane, m0, m1, iv);
        if (lane < 8) P.edgeU[(size_t)r * 8 + lane] = m0 * iv;
        return;
    }
    r -= P.ER;
    if (r == 0) {
        if (lane < 4 && P.L[lane] > 0) {     // mixing weights (kernels.py:402-412), the arithmetic of bank_prepare_task
            const float es = expf(*P.bank[lane].support_attr_sc_weight), ec = expf(*P.bank[lane].center_attr_sc_weight),
                        ee = expf(*P.bank[lane].edge_attr_support_sc_weight);
            const float den = __fadd_rn(__fadd_rn(es, ec), ee);
            const float ws = es / den, wc = ec / den, we = ee / den;
            P.mix[lane * 4 + 0] = ws; P.mix[lane * 4 + 1] = wc; P.mix[lane * 4 + 2] = we;
            P.mix[lane * 4 + 3] = __fadd_rn(__fadd_rn(ws, wc), we);
        }
        return;
    }
    const int t = (r - 1) * 64 + lane;               // chirality table (kernels.py:331-341)
    if (P.bank[3].p_support != nullptr && t < P.L[3] * 12) {
        const int l = t / 12, p = t % 12;
        const float* ps = P.bank[3].p_support + (size_t)l * 12;
        P.chir[t] = (int8_t)triple_sign(ps + 3 * PERM4[p][0], ps + 3 * PERM4[p][1], ps + 3 * PERM4[p][2]);
    }
}

// --------------------------------------------------------------------------------------------- the step ----
// The step kernel's arguments stay in the kernel-argument segment (address space 4: scalar loads, also with a run-time layer
// index); handing the by-value struct to the templated body by reference would make the compiler copy it to scratch.
#define MOL_K __attribute__((address_space(4)))
typedef const MOL_K MolArgs* MolArgsP;
typedef const MOL_K MolLayer MolLayerK;

struct MolMeta {                        // (LDS)
    int deg[MOL_MAXA], nei[MOL_MAXA], rank[MOL_MAXA], mol[MOL_MAXA];
    int dcnt[4]; unsigned char dlist[4][MOL_MAXA];
    float inv[MOL_MAXA]; int big[MOL_MAXA];
    float sgn[MOL_MAXA]; int eq[MOL_MAXA];
    float bn_mu[32], bn_is[32], bn_scale[32], bn_shift[32];
    int mol_first[MOL_MAXM + 1];
    float red[MOL_NW][16];
    float hv[MOL_MAXM][2];
    float mixs[MKGNN_MOLECULE_MAX_LAYERS][16];   // every layer's mixing weights
    const float* p_edge[MKGNN_MOLECULE_MAX_LAYERS];   // unit edge-support rows of a layer, lin1 padded / transposed, lin2: in the
    const float *p_w1p, *p_w1pt, *p_w2;               // LDS cache behind this struct, or where the preparation launch left them
    float mix[16];                      // the current layer's mixing weights [degree][w_s, w_c, w_e, sum] (a global load per use otherwise)
    int back[MOL_MAXA];
    int tgt[5 * 8];                     // backward, degree 4 on the vector pipe: target atom of every part row                 // per atom, 2 bits per slot: the position of the atom in that neighbour's own slot list
    int8_t idx[MOL_MAXA * 112];
};

// pi_p(a) of degree d from immediates: one byte per order, two bits per slot (the tables of kgnn_common.h / kernels.py:109-128)
__device__ __forceinline__ int mol_perm(int d, int p, int a) {
    if (d <= 1) return 0;
    if (d == 2) return a ^ p;
    // byte of order p: slots a = 0..3 in bits 2a
    // d = 3: {0,1,2} {0,2,1} {1,0,2} {1,2,0} {2,0,1} {2,1,0}
    // d = 4: {0,1,2,3} {0,2,3,1} {0,3,1,2} {1,0,3,2} {1,2,0,3} {1,3,2,0} {2,0,1,3} {2,1,3,0} {2,3,0,1} {3,0,2,1} {3,1,0,2} {3,2,1,0}
    constexpr unsigned b3[6] = {0u | 1u << 2 | 2u << 4, 0u | 2u << 2 | 1u << 4, 1u | 0u << 2 | 2u << 4,
                                1u | 2u << 2 | 0u << 4, 2u | 0u << 2 | 1u << 4, 2u | 1u << 2 | 0u << 4};
    constexpr unsigned b4[12] = {0u | 1u << 2 | 2u << 4 | 3u << 6, 0u | 2u << 2 | 3u << 4 | 1u << 6, 0u | 3u << 2 | 1u << 4 | 2u << 6,
                                 1u | 0u << 2 | 3u << 4 | 2u << 6, 1u | 2u << 2 | 0u << 4 | 3u << 6, 1u | 3u << 2 | 2u << 4 | 0u << 6,
                                 2u | 0u << 2 | 1u << 4 | 3u << 6, 2u | 1u << 2 | 3u << 4 | 0u << 6, 2u | 3u << 2 | 0u << 4 | 1u << 6,
                                 3u | 0u << 2 | 2u << 4 | 1u << 6, 3u | 1u << 2 | 0u << 4 | 2u << 6, 3u | 2u << 2 | 1u << 4 | 0u << 6};
    constexpr unsigned w3a = b3[0] | b3[1] << 8 | b3[2] << 16 | b3[3] << 24, w3b = b3[4] | b3[5] << 8;
    constexpr unsigned w4a = b4[0] | b4[1] << 8 | b4[2] << 16 | b4[3] << 24, w4b = b4[4] | b4[5] << 8 | b4[6] << 16 | b4[7] << 24,
                       w4c = b4[8] | b4[9] << 8 | b4[10] << 16 | b4[11] << 24;
    const unsigned w = d == 3 ? (p < 4 ? w3a : w3b) : (p < 4 ? w4a : (p < 8 ? w4b : w4c));
    return (int)((w >> (8 * (p & 3) + 2 * a)) & 3u);
}

__device__ __forceinline__ v4 mfma4(const v4 a, const v4 b, v4 acc) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[c], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ float sigmoid_m(float p) { return 1.f / (1.f + expf(-p)); }

// Philox4x32-10, the head's generator (kgnn_readout.hip): the same (seed, offset, element) gives the same mask as
// mkgnn_bce_head_fused, so the two paths agree on a step with dropout
__device__ __forceinline__ uint32_t mol_philox_word(uint64_t seed, uint64_t offset, uint64_t element) {
    uint32_t c0 = (uint32_t)(element >> 2), c1 = (uint32_t)(element >> 34), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t w[4] = {c0, c1, c2, c3};
    return w[element & 3];
}
__device__ __forceinline__ float mol_keep_scale(uint64_t seed, uint64_t offset, uint64_t element, float p) {
    const float u = (float)(mol_philox_word(seed, offset, element) >> 8) * (1.f / 16777216.f);
    return u >= p ? 1.f / (1.f - p) : 0.f;
}

// 1 / max(|row|, eps) of the NAP rows of `buf` (stride XS, FP columns): eight threads per row
__device__ __forceinline__ void mol_row_norms(const float* buf, int NAP, int XS, int FP, MolMeta& m, int tid) {
    for (int j0 = 0; j0 < NAP; j0 += MOL_THREADS / 8) {
        const int j = j0 + (tid >> 3), s8 = tid & 7;
        float s = 0.f;
        if (j < NAP) for (int f = 4 * s8; f < FP; f += 32) {
            const v4 v = *(const v4*)&buf[j * XS + f];
            s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
        }
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (j < NAP && s8 == 0) { const float nr = sqrtf(s); m.inv[j] = 1.f / fmaxf(nr, MKGNN_EPS); m.big[j] = nr > MKGNN_EPS ? 1 : 0; }
    }
}

// One degree's (atom, kernel) pairs of the forward: permutation scan on the gathered d x d matrix, edge score with the chosen
// order, mix, chirality sign (kernels.py:353-425), written into the chunk's sim rows and the pair records.
template <int D>
__device__ __forceinline__ void mol_pairs_forward(MolLayerK& Y, const float* edgeU, bool last, const float* St, float* sim, const float* bond,
                                                  const MolMeta& m, int tid) {
    const int L = Y.L[D - 1], cnt = m.dcnt[D - 1];
    if (L == 0 || cnt == 0) return;
    const int sup = Y.sup_row[D - 1], cen = Y.cen_row[D - 1], off = Y.off[D - 1];
    const float ws = m.mix[(D - 1) * 4], wc = m.mix[(D - 1) * 4 + 1], we = m.mix[(D - 1) * 4 + 2], wsum = m.mix[(D - 1) * 4 + 3];
    for (int p = tid; p < cnt * L; p += MOL_THREADS) {
        const int ai = p / L, l = p - ai * L;
        const int n = m.dlist[D - 1][ai];
        const int pk = m.nei[n];
        // the kernel's D unit edge-support rows (whichever order wins needs D of them: fetched before the scan)
        v4 es0[D], es1[D];
#pragma unroll
        for (int b = 0; b < D; ++b) {
            const float* es = edgeU + (size_t)(Y.e_row[D - 1] + b * L + l) * 8;
            es0[b] = *(const v4*)es; es1[b] = *(const v4*)(es + 4);
        }
        float cm[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const int j = (pk >> (8 * a)) & 0xFF;
#pragma unroll
            for (int b = 0; b < D; ++b) cm[a][b] = St[j * MOL_RS + sup + b * L + l];
        }
        const float cc = St[n * MOL_RS + cen + l];
        float best; int idx;
        best_permutation<D>(cm, best, idx);
        float ed = 0.f;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const int pb = mol_perm(D, idx, a);
            v4 e0 = es0[0], e1 = es1[0];
#pragma unroll
            for (int b = 1; b < D; ++b) if (pb == b) { e0 = es0[b]; e1 = es1[b]; }
            const v4 b0 = *(const v4*)&bond[(n * 4 + a) * 8], b1 = *(const v4*)&bond[(n * 4 + a) * 8 + 4];
            float dt = b0[0] * e0[0];
            dt = fmaf(b0[1], e0[1], dt); dt = fmaf(b0[2], e0[2], dt); dt = fmaf(b0[3], e0[3], dt);
            dt = fmaf(b1[0], e1[0], dt); dt = fmaf(b1[1], e1[1], dt); dt = fmaf(b1[2], e1[2], dt); dt = fmaf(b1[3], e1[3], dt);
            ed = (a == 0) ? dt : __fadd_rn(ed, dt);
        }
        ed = div_by<D>(ed);
        float sc = __fadd_rn(__fadd_rn(__fmul_rn(best, ws), __fmul_rn(cc, wc)), __fmul_rn(ed, we)) / wsum;
        float ch = 1.f;
        if constexpr (D == 4) {
            if (last && !m.eq[n]) ch = ((float)Y.chir[l * 12 + idx] == m.sgn[n]) ? 1.f : -1.f;
            sc *= ch;
            if (Y.chir_out) Y.chir_out[(size_t)m.rank[n] * L + l] = (int8_t)ch;
        }
        sim[n * MOL_XS + off + l] = sc;
        if (Y.pair[D - 1]) pair_store(Y.pair[D - 1], (size_t)m.rank[n] * L + l, best, cc, ed, idx);
    }
}

// One degree's pairs of the backward: d loss / d sim of the pair from the neighbours' d loss / d h (propagate's adjoint,
// KernelLayer.py:119-123), the three score-weight partials, and the pair's {g, order} into the coefficient tables.
template <int D>
__device__ __forceinline__ v4 mol_pair_prefetch(MolLayerK& Y, const MolMeta& m, int tid) {
    const int L = Y.L[D - 1], cnt = m.dcnt[D - 1];
    if (L == 0 || tid >= cnt * L) return v4{0.f, 0.f, 0.f, 0.f};
    const int ai = tid / L, l = tid - ai * L;
    return pair_load(Y.pair[D - 1], (size_t)m.rank[m.dlist[D - 1][ai]] * L + l);
}

template <int D>
__device__ __forceinline__ void mol_pairs_backward(MolLayerK& Y, bool last, const float* dh, float* gtab, MolMeta& m,
                                                   const v4 first, int tid) {
    const int L = Y.L[D - 1], cnt = m.dcnt[D - 1];
    float ts = 0.f, tc = 0.f, te = 0.f;
    if (L > 0 && cnt > 0) {
        const int off = Y.off[D - 1];
        const float ws = m.mix[(D - 1) * 4], wc = m.mix[(D - 1) * 4 + 1], we = m.mix[(D - 1) * 4 + 2], wsum = m.mix[(D - 1) * 4 + 3];
        for (int p = tid; p < cnt * L; p += MOL_THREADS) {
            const int ai = p / L, l = p - ai * L;
            const int n = m.dlist[D - 1][ai];
            const int pk = m.nei[n];
            float g = dh[((pk) & 0xFF) * MOL_XS + off + l];
#pragma unroll
            for (int a = 1; a < D; ++a) g = __fadd_rn(g, dh[((pk >> (8 * a)) & 0xFF) * MOL_XS + off + l]);
            const size_t o = (size_t)m.rank[n] * L + l;
            const v4 rec = p == tid ? first : pair_load(Y.pair[D - 1], o);
            if constexpr (D == 4) { if (last && Y.chir_out) g *= (float)Y.chir_out[o]; }
            const float scr = __fadd_rn(__fadd_rn(__fmul_rn(rec[0], ws), __fmul_rn(rec[1], wc)), __fmul_rn(rec[2], we)) / wsum;
            ts = fmaf(g, ws * (rec[0] - scr) / wsum, ts);
            tc = fmaf(g, wc * (rec[1] - scr) / wsum, tc);
            te = fmaf(g, we * (rec[2] - scr) / wsum, te);
            gtab[n * MOL_XS + off + l] = g;
            m.idx[n * 112 + off + l] = (int8_t)pair_index(rec);
        }
    }
    // fixed-order reduction: xor tree inside a wave, the eight waves in order
    ts = wave_sum(ts); tc = wave_sum(tc); te = wave_sum(te);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) { m.red[wave][(D - 1) * 3] = ts; m.red[wave][(D - 1) * 3 + 1] = tc; m.red[wave][(D - 1) * 3 + 2] = te; }
}

// Cf entries of one degree's bank rows for every atom j of the chunk (one owner thread per entry: fixed order, no atomics):
// support (b, l): sum over j's neighbours n of degree D whose chosen order maps the slot that points at j to b, of
// g(n, l) w_s / (W D); centre l: g(j, l) w_c / W if j itself has degree D.
template <int D>
__device__ __forceinline__ void mol_cf_degree(MolLayerK& Y, int NR, const float* gtab, float* Cf, const MolMeta& m, int tid) {
    const int L = Y.L[D - 1];
    if (L == 0) return;
    const int off = Y.off[D - 1], sup = Y.sup_row[D - 1], cen = Y.cen_row[D - 1];
    const float wsd = m.mix[(D - 1) * 4] / m.mix[(D - 1) * 4 + 3] / (float)D, wcd = m.mix[(D - 1) * 4 + 1] / m.mix[(D - 1) * 4 + 3];
    for (int it = tid; it < NR * L; it += MOL_THREADS) {
        const int j = it / L, l = it - j * L;
        const int dj = m.deg[j], pkj = m.nei[j], bkj = m.back[j];
        float acc[D];
#pragma unroll
        for (int b = 0; b < D; ++b) acc[b] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            if (s2 >= dj) continue;
            const int n = (pkj >> (8 * s2)) & 0xFF;
            if (m.deg[n] != D) continue;
            const int b = mol_perm(D, m.idx[n * 112 + off + l], (bkj >> (2 * s2)) & 3);
            const float gv = gtab[n * MOL_XS + off + l] * wsd;
#pragma unroll
            for (int bb = 0; bb < D; ++bb) acc[bb] += bb == b ? gv : 0.f;
        }
#pragma unroll
        for (int b = 0; b < D; ++b) Cf[j * MOL_RS + sup + b * L + l] = acc[b];
        Cf[j * MOL_RS + cen + l] = dj == D ? gtab[j * MOL_XS + off + l] * wcd : 0.f;
    }
}

// Gradient of one degree's unit edge-support rows: Be[(b, l)] = sum over the degree's atoms (ascending) of g_e * the unit bond
// row of the slot the chosen order maps to b; half a row (16 bytes) per thread.
template <int D>
__device__ __forceinline__ void mol_edge_grad_degree(MolLayerK& Y, const float* gtab, const float* bond, float* eslab, const MolMeta& m, int tid) {
    const int L = Y.L[D - 1], cnt = m.dcnt[D - 1];
    if (L == 0) return;
    const int off = Y.off[D - 1];
    const float we = m.mix[(D - 1) * 4 + 2] / m.mix[(D - 1) * 4 + 3] / (float)D;
    for (int it = tid; it < D * L * 2; it += MOL_THREADS) {
        const int row = it >> 1, half = it & 1, b = row / L, l = row - b * L;
        v4 acc = v4{0.f, 0.f, 0.f, 0.f};
        for (int ai = 0; ai < cnt; ++ai) {
            const int n = m.dlist[D - 1][ai];
            const int idx = m.idx[n * 112 + off + l];
            int sl = 0;
#pragma unroll
            for (int s = 1; s < D; ++s) if (mol_perm(D, idx, s) == b) sl = s;
            const float ge = gtab[n * MOL_XS + off + l] * we;
            const v4 bv = *(const v4*)&bond[(n * 4 + sl) * 8 + 4 * half];
            acc[0] = fmaf(ge, bv[0], acc[0]); acc[1] = fmaf(ge, bv[1], acc[1]); acc[2] = fmaf(ge, bv[2], acc[2]); acc[3] = fmaf(ge, bv[3], acc[3]);
        }
        *(v4*)&eslab[(size_t)(Y.e_row[D - 1] + row) * 8 + 4 * half] = acc;
    }
}

// ---- Degree 4 without the dense pass.  Its bank is more than half of a layer's rows (5 L_4 of them) while a few per cent of
// the atoms have four bonds: multiplying every atom of the chunk with those rows is the largest single waste of the dense
// form.  With at most MOL_D4_VALU_MAX such atoms in the chunk their 17 dot products per kernel run on the vector pipe, eight
// lanes per (atom, kernel) pair, each lane a strided eighth of the feature chunks; the backward likewise touches only the
// rows and atoms involved.  More degree-4 atoms than that: the dense pass 1.
constexpr int MOL_D4_VALU_MAX = 6;

__device__ __forceinline__ float mol_dot4(const v4 a, const v4 b, float acc) {
    acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc); acc = fmaf(a[2], b[2], acc); return fmaf(a[3], b[3], acc);
}

__device__ __forceinline__ void mol_pairs4_valu(MolLayerK& Y, const float* edgeU, bool last, const float* xin, int XS, float* sim, const float* bond,
                                                const MolMeta& m, int tid) {
    constexpr int D = 4;
    const int L = Y.L[3], cnt = m.dcnt[3];
    if (L == 0 || cnt == 0) return;
    const int FP = Y.FP, KV = FP >> 2, off = Y.off[3];
    const float* sup0 = Y.bankU + (size_t)(Y.row_base[1] + Y.sup_row[3]) * FP;
    const float* cen0 = Y.bankU + (size_t)(Y.row_base[1] + Y.cen_row[3]) * FP;
    const float ws = m.mix[12], wc = m.mix[13], we = m.mix[14], wsum = m.mix[15];
    for (int it = tid; it < cnt * L * 8; it += MOL_THREADS) {
        const int s8 = it & 7, p = it >> 3, ai = p / L, l = p - ai * L;
        const int n = m.dlist[3][ai], pk = m.nei[n];
        int nb[D];
#pragma unroll
        for (int a = 0; a < D; ++a) nb[a] = (pk >> (8 * a)) & 0xFF;
        v4 es0[D], es1[D];
#pragma unroll
        for (int b = 0; b < D; ++b) {
            const float* es = edgeU + (size_t)(Y.e_row[3] + b * L + l) * 8;
            es0[b] = *(const v4*)es; es1[b] = *(const v4*)(es + 4);
        }
        float cm[D][D], cc = 0.f;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) cm[a][b] = 0.f;
        for (int k = s8; k < KV; k += 8) {
            v4 sb[D];
#pragma unroll
            for (int b = 0; b < D; ++b) sb[b] = *(const v4*)(sup0 + (size_t)(b * L + l) * FP + 4 * k);
            const v4 sc = *(const v4*)(cen0 + (size_t)l * FP + 4 * k);
            cc = mol_dot4(*(const v4*)&xin[n * XS + 4 * k], sc, cc);
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const v4 xa = *(const v4*)&xin[nb[a] * XS + 4 * k];
#pragma unroll
                for (int b = 0; b < D; ++b) cm[a][b] = mol_dot4(xa, sb[b], cm[a][b]);
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            cc += __shfl_xor(cc, o, 64);
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b < D; ++b) cm[a][b] += __shfl_xor(cm[a][b], o, 64);
        }
        cc *= m.inv[n];
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const float iv = m.inv[nb[a]];
#pragma unroll
            for (int b = 0; b < D; ++b) cm[a][b] *= iv;
        }
        float best; int idx;
        best_permutation<D>(cm, best, idx);
        float ed = 0.f;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const int pb = mol_perm(D, idx, a);
            v4 e0 = es0[0], e1 = es1[0];
#pragma unroll
            for (int b = 1; b < D; ++b) if (pb == b) { e0 = es0[b]; e1 = es1[b]; }
            const v4 b0 = *(const v4*)&bond[(n * 4 + a) * 8], b1 = *(const v4*)&bond[(n * 4 + a) * 8 + 4];
            float dt = b0[0] * e0[0];
            dt = fmaf(b0[1], e0[1], dt); dt = fmaf(b0[2], e0[2], dt); dt = fmaf(b0[3], e0[3], dt);
            dt = fmaf(b1[0], e1[0], dt); dt = fmaf(b1[1], e1[1], dt); dt = fmaf(b1[2], e1[2], dt); dt = fmaf(b1[3], e1[3], dt);
            ed = (a == 0) ? dt : __fadd_rn(ed, dt);
        }
        ed = div_by<D>(ed);
        float sc = __fadd_rn(__fadd_rn(__fmul_rn(best, ws), __fmul_rn(cc, wc)), __fmul_rn(ed, we)) / wsum;
        float ch = 1.f;
        if (last && !m.eq[n]) ch = ((float)Y.chir[l * 12 + idx] == m.sgn[n]) ? 1.f : -1.f;
        sc *= ch;
        if (s8 == 0) {
            if (Y.chir_out) Y.chir_out[(size_t)m.rank[n] * L + l] = (int8_t)ch;
            sim[n * MOL_XS + off + l] = sc;
            if (Y.pair[3]) pair_store(Y.pair[3], (size_t)m.rank[n] * L + l, best, cc, ed, idx);
        }
    }
}

// Backward of the same: (A) for every degree-4 atom the rows  sum_l coef(n, l) * unit bank row  that go to each of its four
// neighbours and to itself (`part`: [atoms][5][XS], one thread per 16-byte piece, kernels in order; merged into G afterwards
// in part order);
// (B) the gradient of the degree-4 bank rows, straight into the chunk's slab (every row / piece has one owner thread, the
// degree-4 atoms in ascending order).
__device__ __forceinline__ void mol_bwd4_rows(MolLayerK& Y, const float* gtab, float* part, MolMeta& m, int tid) {
    const int L = Y.L[3], cnt = m.dcnt[3], FP = Y.FP, KV = FP >> 2, XS = FP + 4, off = Y.off[3];
    if (tid < cnt * 5) {                                  // whose row each part is: neighbour t of the atom, or the atom itself
        const int ia = tid / 5, t = tid - ia * 5, n = m.dlist[3][ia];
        m.tgt[tid] = t < 4 ? (m.nei[n] >> (8 * t)) & 0xFF : n;
    }
    const float* sup0 = Y.bankU + (size_t)(Y.row_base[1] + Y.sup_row[3]) * FP;
    const float* cen0 = Y.bankU + (size_t)(Y.row_base[1] + Y.cen_row[3]) * FP;
    const float wsd = m.mix[12] / m.mix[15] * 0.25f, wcd = m.mix[13] / m.mix[15];
    // eight lanes per 16-byte piece of a part row, each taking every eighth kernel (one round of loads in flight per lane),
    // combined by a fixed xor tree
    for (int it = tid; it < cnt * 5 * KV * 8; it += MOL_THREADS) {
        const int s8 = it & 7, pc = it >> 3, k = pc % KV, t = (pc / KV) % 5, ia = pc / (5 * KV);
        const int n = m.dlist[3][ia];
        v4 acc = v4{0.f, 0.f, 0.f, 0.f};
        for (int l0 = s8; l0 < L; l0 += 64) {
            v4 bv[8]; float c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lw = l0 + 8 * u, l = lw < L ? lw : L - 1;
                const float g = gtab[n * MOL_XS + off + l];
                const float* row = t < 4 ? sup0 + (size_t)(mol_perm(4, m.idx[n * 112 + off + l], t) * L + l) * FP : cen0 + (size_t)l * FP;
                bv[u] = *(const v4*)(row + 4 * k);
                c[u] = lw < L ? g * (t < 4 ? wsd : wcd) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[0] = fmaf(c[u], bv[u][0], acc[0]); acc[1] = fmaf(c[u], bv[u][1], acc[1]);
                acc[2] = fmaf(c[u], bv[u][2], acc[2]); acc[3] = fmaf(c[u], bv[u][3], acc[3]);
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
        if (s8 == 0) *(v4*)&part[(ia * 5 + t) * XS + 4 * k] = acc;
    }
}

__device__ __forceinline__ void mol_bwd4_bank(MolLayerK& Y, const float* gtab, const float* U, float* bslab, const MolMeta& m, int tid) {
    const int L = Y.L[3], cnt = m.dcnt[3], FP = Y.FP, KV = FP >> 2, XS = FP + 4, off = Y.off[3];
    const float wsd = m.mix[12] / m.mix[15] * 0.25f, wcd = m.mix[13] / m.mix[15];
    const int sup = Y.sup_row[3], cen = Y.cen_row[3];
    for (int it = tid; it < 5 * L * KV; it += MOL_THREADS) {
        const int k = it % KV, row = it / KV;             // row < 4 L: support (b, l); else centre l
        const int b = row < 4 * L ? row / L : 4, l = row - b * L;
        v4 acc = v4{0.f, 0.f, 0.f, 0.f};
        for (int ai = 0; ai < cnt; ++ai) {
            const int n = m.dlist[3][ai];
            const float g = gtab[n * MOL_XS + off + l];
            int src = n;
            if (b < 4) {
                const int idx = m.idx[n * 112 + off + l], pk = m.nei[n];
                int sl = 0;
#pragma unroll
                for (int s = 1; s < 4; ++s) if (mol_perm(4, idx, s) == b) sl = s;
                src = (pk >> (8 * sl)) & 0xFF;
            }
            const v4 uv = *(const v4*)&U[src * XS + 4 * k];
            const float c = g * (b < 4 ? wsd : wcd);
            acc[0] = fmaf(c, uv[0], acc[0]); acc[1] = fmaf(c, uv[1], acc[1]); acc[2] = fmaf(c, uv[2], acc[2]); acc[3] = fmaf(c, uv[3], acc[3]);
        }
        *(v4*)&bslab[(size_t)((b < 4 ? sup + b * L : cen) + l) * FP + 4 * k] = acc;
    }
}

// NT: atom tiles the LDS buffers are laid out for (the launch's largest chunk); NTR <= NT: atom tiles of THIS chunk, a
// compile-time count so that every product loop is straight-line code
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding vector-memory operation
// (s_waitcnt vmcnt(0)): the stores of a phase (pair records, kept input rows, slab rows) and the loads issued a phase ahead
// would be drained at each of the ~70 barriers of a chunk.  Nothing the kernel writes to global memory is read back before
// the one full barrier at the forward -> backward transition.
#define MOL_BAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int NT, int NTR>
__device__ __forceinline__ void molecule_step_body(MolArgsP ap, float* lds) {
    const MOL_K MolArgs& a = *ap;
    constexpr int NAP = 16 * NT;
    float* bufA = lds;                                   // [NAP, XS]
    float* bufB = bufA + NAP * MOL_XS;                   // [NAP, XS]
    float* St = bufB + NAP * MOL_XS;                     // [NAP, RS]  (Cf in the backward; readout scratch)
    float* bond = St + NAP * MOL_RS;                     // [NAP, 4, 8] unit bond rows of every atom's slots
    MolMeta& m = *(MolMeta*)(bond + NAP * 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int chunk = blockIdx.x;
    const int m0 = a.chunk_ptr[chunk], m1 = a.chunk_ptr[chunk + 1], nm = m1 - m0;
    const int64_t a0 = a.mol_ptr[m0];
    const int NA = (int)(a.mol_ptr[m1] - a0);
    constexpr int ntr = NTR, NR = 16 * NTR;              // atom tiles / rows this chunk has
    constexpr int ksa = 4 * NTR;                         // k-steps of a product that contracts over the chunk's atoms
    float* slab = a.slab + (size_t)chunk * a.slab_floats;
    const bool do_head = (a.mode & MKGNN_MOLECULE_HEAD) != 0, do_bwd = (a.mode & MKGNN_MOLECULE_BACKWARD) != 0;
    {
        float* cache = (float*)((char*)&m + ((sizeof(MolMeta) + 15) & ~(size_t)15));
        if (tid < a.nl * 16) m.mixs[tid >> 4][tid & 15] = a.layer[tid >> 4].mix[tid & 15];
        if (a.cache_on) {
            for (int li = 0; li < a.nl; ++li) {
                const v4* src = (const v4*)a.layer[li].edgeU;
                v4* dst = (v4*)(cache + a.c_edge[li]);
                for (int i = tid; i < a.layer[li].ER * 2; i += MOL_THREADS) dst[i] = src[i];
            }
            for (int i = tid; i < a.HP * 28; i += MOL_THREADS) { ((v4*)(cache + a.c_w1p))[i] = ((const v4*)a.w1p)[i]; ((v4*)(cache + a.c_w1pt))[i] = ((const v4*)a.w1pt)[i]; }
            for (int i = tid; i < a.G * a.H; i += MOL_THREADS) cache[a.c_w2 + i] = a.w2[i];
        }
        if (tid < a.nl) m.p_edge[tid] = a.cache_on ? cache + a.c_edge[tid] : a.layer[tid].edgeU;
        if (tid == 0) {
            m.p_w1p = a.cache_on ? cache + a.c_w1p : a.w1p;
            m.p_w1pt = a.cache_on ? cache + a.c_w1pt : a.w1pt;
            m.p_w2 = a.cache_on ? cache + a.c_w2 : a.w2;
        }
    }
    int stamp_i = 0;
#define MOL_STAMP() do { if (a.stamps && chunk == 0 && tid == 0) a.stamps[stamp_i] = __builtin_readcyclecounter(); ++stamp_i; } while (0)
    MOL_STAMP();

    // ------------------------------------------------------------------------------------------ chunk metadata ----
    if (tid <= nm) m.mol_first[tid] = (int)(a.mol_ptr[m0 + tid] - a0);
    if (tid < MOL_MAXA) {
        int dg = 0, pk = 0, rk = 0;
        if (tid < NA) {
            dg = a.atom_deg[a0 + tid];
            rk = a.atom_rank[a0 + tid];
            if (dg >= 1 && dg <= 4) {
                const int64_t* nb = a.nei[dg - 1] + (size_t)rk * dg;
                for (int s = 0; s < dg; ++s) pk |= ((int)(nb[s] - a0) & 0xFF) << (8 * s);
            } else dg = 0;
        }
        if (tid < NAP) { m.deg[tid] = dg; m.nei[tid] = pk; m.rank[tid] = rk; }
        // the atoms of every degree, ascending (one wave: ballots)
#pragma unroll
        for (int d = 1; d <= 4; ++d) {
            const unsigned long long mask = __ballot(dg == d);
            if (dg == d) m.dlist[d - 1][__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned char)tid;
            if (tid == 0) m.dcnt[d - 1] = __popcll(mask);
        }
        // sign of the neighbour tetrahedron of a degree-4 atom (kernels.py:327-337)
        float sg = 0.f;
        if (dg == 4 && a.p_focal4 && a.nei_p4) {
            float t[3][3];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int c = 0; c < 3; ++c) t[s][c] = __fsub_rn(a.nei_p4[((size_t)rk * 4 + s) * 3 + c], a.p_focal4[(size_t)rk * 3 + c]);
            sg = triple_sign(t[0], t[1], t[2]);
        }
        if (tid < NAP) { m.sgn[tid] = sg; m.eq[tid] = 0; }
    }
    MOL_BAR();
    // molecule of every atom; unit bond rows of every slot
    if (tid < NAP) {
        int g = 0;
        while (g + 1 < nm && tid >= m.mol_first[g + 1]) ++g;
        m.mol[tid] = g;
        int bk = 0;
        const int dj = m.deg[tid], pkj = m.nei[tid];
        for (int s2 = 0; s2 < dj; ++s2) {
            const int n = (pkj >> (8 * s2)) & 0xFF, pkn = m.nei[n], dn = m.deg[n];
            int sl = 0;
            for (int s3 = 0; s3 < dn; ++s3) if (((pkn >> (8 * s3)) & 0xFF) == tid) sl = s3;
            bk |= sl << (2 * s2);
        }
        m.back[tid] = bk;
    }
    for (int it = tid; it < NAP * 8; it += MOL_THREADS) {
        const int j = it >> 3, s = (it >> 1) & 3, half = it & 1;
        const int dg = m.deg[j];
        v4 v = v4{0.f, 0.f, 0.f, 0.f};
        if (s < dg) v = *(const v4*)(a.eunit[dg - 1] + ((size_t)m.rank[j] * dg + s) * 8 + 4 * half);
        *(v4*)&bond[(j * 4 + s) * 8 + 4 * half] = v;
    }
    MOL_STAMP();    // 1: metadata
    // ------------------------------------------------------------------------------------------------ batch norm ----
    const int C = a.F0;
    {
        // mean = sum of the blocks' column sums / N;  M2 = sum of the blocks' centred squares + n_b (mean_b - mean)^2
        // (16 threads per column take every 16th block; combined in a fixed order)
        float* red = St;                                  // [16][32] scratch
        const int c = tid & 31, k = tid >> 5, cc = c < C ? c : 0;
        const int64_t per = (a.n_atoms + a.bn_nblk - 1) / a.bn_nblk;
        float part = 0.f;
        if (a.bn_training && k < 16) for (int b = k; b < a.bn_nblk; b += 16) part += a.bn_part[(size_t)b * 2 * C + cc];
        if (k < 16) red[k * 32 + c] = part;
        MOL_BAR();
        float mu = 0.f;
        if (a.bn_training) {
            for (int kk = 0; kk < 16; ++kk) mu += red[kk * 32 + c];
            mu /= (float)a.n_atoms;
        }
        MOL_BAR();
        part = 0.f;
        if (a.bn_training && k < 16) for (int b = k; b < a.bn_nblk; b += 16) {
            const int64_t lo = per * b, hi = lo + per < a.n_atoms ? lo + per : a.n_atoms;
            if (hi > lo) {
                const float nb = (float)(hi - lo), dm = a.bn_part[(size_t)b * 2 * C + cc] / nb - mu;
                part += a.bn_part[(size_t)b * 2 * C + C + cc] + nb * dm * dm;
            }
        }
        if (k < 16) red[k * 32 + c] = part;
        MOL_BAR();
        if (tid < 32) {
            float is = 1.f;
            if (tid < C) {
                if (a.bn_training) {
                    float m2 = 0.f;
                    for (int kk = 0; kk < 16; ++kk) m2 += red[kk * 32 + tid];
                    const float var = m2 / (float)a.n_atoms;
                    is = 1.f / sqrtf(var + a.bn_eps);
                    if (chunk == 0) {
                        if (a.run_mean) a.run_mean[tid] = fmaf(a.bn_mom, mu - a.run_mean[tid], a.run_mean[tid]);
                        if (a.run_var) {
                            const float unb = a.n_atoms > 1 ? var * ((float)a.n_atoms / (float)(a.n_atoms - 1)) : var;
                            a.run_var[tid] = fmaf(a.bn_mom, unb - a.run_var[tid], a.run_var[tid]);
                        }
                    }
                } else {
                    mu = a.run_mean[tid];
                    is = 1.f / sqrtf(a.run_var[tid] + a.bn_eps);
                }
            } else mu = 0.f;
            m.bn_mu[tid] = mu; m.bn_is[tid] = is;
            m.bn_scale[tid] = tid < C ? is * (a.bn_w ? a.bn_w[tid] : 1.f) : 0.f;
            m.bn_shift[tid] = tid < C ? (a.bn_b ? a.bn_b[tid] : 0.f) : 0.f;
            if (chunk == 0 && tid == 0 && a.nbt && a.bn_training) a.nbt[0] += 1;
        }
    }
    MOL_BAR();
    const int XS0 = a.layer[0].FP + 4;
    for (int it = tid; it < NR * 32; it += MOL_THREADS) {
        const int j = it >> 5, c = it & 31;
        float v = 0.f;
        if (j < NA && c < C) v = fmaf(a.x[(a0 + j) * a.xs + c] - m.bn_mu[c], m.bn_scale[c], m.bn_shift[c]);
        if (c < a.layer[0].FP) bufA[j * XS0 + c] = v;
    }
    MOL_BAR();

    MOL_STAMP();    // 2: batch norm
    // ---------------------------------------------------------------------------------------------- the layers ----
    float* xin = bufA;
    float* sim = bufB;
    for (int li = 0; li < a.nl; ++li) {
        MolLayerK& Y = a.layer[li];
        const bool last = li == a.nl - 1;
        const int FP = Y.FP, XS = FP + 4, KJ = Y.KJ;
        // the unit bank rows of this wave's (at most two) row tiles of a pass: fetched a phase ahead of their products (a load
        // from L2 behind every tile's products would be an exposed round trip per tile)
        v4 bvn[MOL_TPW][7];
        auto fetch_pass = [&](int p) {
            const int RT = Y.pass_rows[p] >> 4;
#pragma unroll
            for (int t = 0; t < MOL_TPW; ++t) {
                const int nt = wave + MOL_NW * t;
                const float* brow = Y.bankU + (size_t)(Y.row_base[p] + (nt < RT ? nt : 0) * 16 + r) * FP + 4 * q;
#pragma unroll
                for (int j = 0; j < 7; ++j) bvn[t][j] = (j < KJ && nt < RT) ? *(const v4*)(brow + 16 * j) : v4{0.f, 0.f, 0.f, 0.f};
            }
        };
        fetch_pass(0);
        const float* edgeU = m.p_edge[li];
        if (tid < 16) m.mix[tid] = m.mixs[li][tid];
        mol_row_norms(xin, NR, XS, FP, m, tid);
        // this layer's input rows, kept for the backward half (and zero the sim rows)
        if (do_bwd) for (int it = tid; it < NA * (FP / 4); it += MOL_THREADS) {
            const int j = it / (FP / 4), c4 = it - j * (FP / 4);
            *(v4*)(Y.x_save + (size_t)(a0 + j) * FP + 4 * c4) = *(const v4*)&xin[j * XS + 4 * c4];
        }
        for (int it = tid; it < NR * (MOL_XS / 4); it += MOL_THREADS) *(v4*)&sim[4 * it] = v4{0.f, 0.f, 0.f, 0.f};
        // degree 4, last layer: are two of the four neighbour rows bit-identical?  (kernels.py:310-317; a half-wave per atom)
        if (last && m.dcnt[3] > 0) {
            for (int ai = tid >> 5; ai < m.dcnt[3]; ai += MOL_THREADS / 32) {
                const int n = m.dlist[3][ai], pk = m.nei[n], h32 = tid & 31;
                unsigned diff = 0;
                for (int f = h32; f < FP; f += 32) {
                    float v[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[s] = xin[((pk >> (8 * s)) & 0xFF) * XS + f];
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jx = i + 1; jx < 4; ++jx, ++k) if (!(v[i] == v[jx])) diff |= 1u << k;
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) diff |= (unsigned)__shfl_xor((int)diff, o, 64);
                if (h32 == 0) m.eq[n] = diff != 0x3Fu ? 1 : 0;
            }
        }
        MOL_BAR();
        MOL_STAMP();   // layer: norms, save, equal-row test
        // (A operand: the chunk's rows, k-permuted 16-byte reads -- lane (r, q) holds columns 16 j + 4 q .. of row r; read per
        // product tile: seven LDS reads against 28 matrix instructions)
        const bool d4_valu = m.dcnt[3] <= MOL_D4_VALU_MAX;      // (chunk-uniform)
        for (int p = 0; p < 2; ++p) {
            const int RT = Y.pass_rows[p] >> 4;
            if (p == 1 && d4_valu) { MOL_STAMP(); MOL_STAMP(); continue; }
            if (RT == 0) continue;
#pragma unroll
            for (int t = 0; t < MOL_TPW; ++t) {
                const int nt = wave + MOL_NW * t;
                if (nt >= RT) continue;
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    if (mt >= ntr) continue;
                    v4 av[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) av[j] = j < KJ ? *(const v4*)&xin[(mt * 16 + r) * XS + 16 * j + 4 * q] : v4{0.f, 0.f, 0.f, 0.f};
                    v4 acc = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 7; ++j) if (j < KJ) acc = mfma4(av[j], bvn[t][j], acc);
#pragma unroll
                    for (int i = 0; i < 4; ++i) St[(mt * 16 + 4 * q + i) * MOL_RS + nt * 16 + r] = acc[i] * m.inv[mt * 16 + 4 * q + i];
                }
            }
            if (p == 0 && !d4_valu) fetch_pass(1);      // (arrives under the pairs of pass 0)
            MOL_BAR();
            MOL_STAMP();   // layer: products of the pass
            if (p == 0) { mol_pairs_forward<1>(Y, edgeU, last, St, sim, bond, m, tid); mol_pairs_forward<2>(Y, edgeU, last, St, sim, bond, m, tid);
                          mol_pairs_forward<3>(Y, edgeU, last, St, sim, bond, m, tid);
                          if (d4_valu) mol_pairs4_valu(Y, edgeU, last, xin, XS, sim, bond, m, tid); }
            else mol_pairs_forward<4>(Y, edgeU, last, St, sim, bond, m, tid);
            MOL_BAR();
            MOL_STAMP();   // layer: pairs of the pass
        }
        if (Y.sim_out) for (int it = tid; it < NA * Y.K; it += MOL_THREADS) {
            const int j = it / Y.K, c = it - j * Y.K;
            Y.sim_out[(size_t)(a0 + j) * Y.sim_stride + c] = sim[j * MOL_XS + c];
        }
        // propagate: h[i] = sum over i's bonds of sim[neighbour] (KernelLayer.py:119-123; slot order), into the old input buffer
        for (int it = tid; it < NR * 28; it += MOL_THREADS) {
            const int i = it / 28, c4 = it - i * 28;
            const int dg = m.deg[i], pk = m.nei[i];
            v4 s = v4{0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < dg; ++sl) {
                const v4 v = *(const v4*)&sim[((pk >> (8 * sl)) & 0xFF) * MOL_XS + 4 * c4];
                s = sl == 0 ? v : s + v;
            }
            *(v4*)&xin[i * MOL_XS + 4 * c4] = s;
        }
        MOL_BAR();
        MOL_STAMP();   // layer: propagate
    }
    // xin = h of the last layer [NAP, 116]; sim free; St free
    // -------------------------------------------------------------------------------------------------- readout ----
    const int H = a.H, G = a.G, HP = a.HP, HT = HP >> 4;
    float* pre = St;                                     // [NAP, HS]  lin1 output (+ bias); later d loss / d pre
    float* pooled = St + NAP * MOL_HS;                   // [MAXM, 64]
    float* embs = pooled + MOL_MAXM * 64;                // [MAXM, 64]
    float* dembs = embs + MOL_MAXM * 64;                 // [MAXM, 64]
    float* dpool = dembs + MOL_MAXM * 64;                // [MAXM, 64]
    {
        for (int t = wave; t < ntr * HT; t += MOL_NW) {
            const int mt = t / HT, ht = t - mt * HT;
            v4 acc = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const v4 av = *(const v4*)&xin[(mt * 16 + r) * MOL_XS + 16 * j + 4 * q];
                const v4 bv = *(const v4*)(m.p_w1p + (size_t)(ht * 16 + r) * 112 + 16 * j + 4 * q);
                acc = mfma4(av, bv, acc);
            }
            const float bias = (a.b1 && ht * 16 + r < H) ? a.b1[ht * 16 + r] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[(mt * 16 + 4 * q + i) * MOL_HS + ht * 16 + r] = acc[i] + bias;
        }
        MOL_BAR();
        // the molecules' sums of swish(pre): 8 threads per (molecule, hidden unit)
        for (int it = tid; it < nm * H * 8; it += MOL_THREADS) {
            const int s8 = it & 7, gh = it >> 3, g = gh / H, h = gh - g * H;
            float s = 0.f;
            for (int j = m.mol_first[g] + s8; j < m.mol_first[g + 1]; j += 8) { const float x = pre[j * MOL_HS + h]; s += x * sigmoid_m(x); }
            s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            if (s8 == 0) pooled[g * 64 + h] = s;
        }
        MOL_BAR();
        for (int it = tid; it < nm * G * 8; it += MOL_THREADS) {
            const int s8 = it & 7, go = it >> 3, g = go / G, o = go - g * G;
            float s = 0.f;
            for (int h = s8; h < H; h += 8) s = fmaf(m.p_w2[(size_t)o * H + h], pooled[g * 64 + h], s);
            s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            if (s8 == 0) {
                if (a.b2) s = fmaf((float)(m.mol_first[g + 1] - m.mol_first[g]), a.b2[o], s);
                embs[g * 64 + o] = s;
                a.emb[(size_t)(m0 + g) * G + o] = s;
            }
        }
        MOL_BAR();
    }
    float* small = slab;
    MOL_STAMP();   // readout forward
    // ----------------------------------------------------------------------------------------------------- head ----
    if (do_head) {
        // dropout -> ffn -> BCE with logits (model.py:150, 169, 190-198); a half-wave per molecule
        const int g = tid >> 5, h32 = tid & 31;
        const bool drop = a.head_drop > 0.f;
        const uint64_t seed = drop ? (uint64_t)a.rng[0] : 0, offset = drop ? (uint64_t)a.rng[1] : 0;
        float x = 0.f, ks[2] = {1.f, 1.f};
        if (g < nm) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int o = h32 + 32 * k;
                if (o < G) {
                    if (drop) ks[k] = mol_keep_scale(seed, offset, (uint64_t)(m0 + g) * G + o, a.head_drop);
                    x = fmaf(embs[g * 64 + o] * ks[k], a.ffn_w[o], x);
                }
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        float dv = 0.f, ls = 0.f;
        if (g < nm) {
            x += a.ffn_b ? a.ffn_b[0] : 0.f;
            const float yv = a.y[m0 + g];
            dv = (sigmoid_m(x) - yv) / (float)a.n_mols;
            ls = fmaxf(x, 0.f) - x * yv + log1pf(expf(-fabsf(x)));
            if (h32 == 0) a.pred[m0 + g] = x;
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int o = h32 + 32 * k; if (o < G) dembs[g * 64 + o] = dv * a.ffn_w[o] * ks[k]; }
        }
        // per-chunk partials: loss, d bias, d weight (molecule order)
        if (h32 == 0 && g < nm) { m.hv[g][0] = dv; m.hv[g][1] = ls; }
        if (g < nm) {
#pragma unroll
            for (int k = 0; k < 2; ++k) { const int o = h32 + 32 * k; if (o < G) dpool[g * 64 + o] = dv * embs[g * 64 + o] * ks[k]; }
        }
        MOL_BAR();
        if (tid < G) {
            float s = 0.f;
            for (int gg = 0; gg < nm; ++gg) s += dpool[gg * 64 + tid];
            small[a.s_ffn + tid] = s;                           // d ffn weight
        } else if (tid < G + 2) {
            float s = 0.f;
            for (int gg = 0; gg < nm; ++gg) s += m.hv[gg][tid - G];
            small[a.s_loss + (tid - G == 0 ? 1 : 0)] = s;       // [s_loss] sum of the molecules' losses, [s_loss + 1] d ffn bias
        }
        MOL_BAR();
    } else if (do_bwd) {
        for (int it = tid; it < nm * G; it += MOL_THREADS) {
            const int g = it / G, o = it - g * G;
            dembs[g * 64 + o] = a.demb[(size_t)(m0 + g) * G + o];
        }
        MOL_BAR();
    }
    MOL_STAMP();   // head
    if (!do_bwd) return;
    __threadfence_block();      // (pair records, chirality signs and input rows written above are read back below by other threads)
    __syncthreads();

    // ------------------------------------------------------------------------------------- readout, backward ----
    for (int it = tid; it < nm * H * 8; it += MOL_THREADS) {     // d pooled = d emb . W2   (8 threads per element)
        const int s8 = it & 7, gh = it >> 3, g = gh / H, h = gh - g * H;
        float s = 0.f;
        for (int o = s8; o < G; o += 8) s = fmaf(dembs[g * 64 + o], m.p_w2[(size_t)o * H + h], s);
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (s8 == 0) dpool[g * 64 + h] = s;
    }
    for (int it = tid; it < G * H + G; it += MOL_THREADS) {      // d W2 [G, H], d b2 [G]  (partials of this chunk)
        if (it < G * H) {
            const int o = it / H, h = it - o * H;
            float s = 0.f;
            for (int g = 0; g < nm; ++g) s = fmaf(dembs[g * 64 + o], pooled[g * 64 + h], s);
            small[a.s_lin2 + it] = s;
        } else {
            const int o = it - G * H;
            float s = 0.f;
            for (int g = 0; g < nm; ++g) s = fmaf(dembs[g * 64 + o], (float)(m.mol_first[g + 1] - m.mol_first[g]), s);
            small[a.s_lin2 + it] = s;
        }
    }
    MOL_BAR();
    for (int it = tid; it < NR * HP; it += MOL_THREADS) {        // d pre = d pooled[mol] * swish'(pre), in place
        const int j = it / HP, h = it - j * HP;
        float v = 0.f;
        if (j < NA && h < H) {
            const float x = pre[j * MOL_HS + h], sg = sigmoid_m(x);
            v = dpool[m.mol[j] * 64 + h] * (sg * (1.f + x * (1.f - sg)));
        }
        pre[j * MOL_HS + h] = v;
    }
    MOL_BAR();
    for (int it = tid; it < H * 8; it += MOL_THREADS) {          // d b1   (8 threads per hidden unit)
        const int s8 = it & 7, h = it >> 3;
        float s = 0.f;
        for (int j = s8; j < NA; j += 8) s += pre[j * MOL_HS + h];
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (s8 == 0) small[a.s_lin1b + h] = s;
    }
    {
        // d W1 [HP, 112] = d pre^T . h   (this chunk's partial, straight from the accumulators)
        float* dw1 = slab + a.s_dw1;
        for (int t = wave; t < HT * 7; t += MOL_NW) {
            const int ht = t / 7, ft = t - ht * 7;
            v4 acc = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4 * NT; ++s)
                if (s < ksa) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pre[(4 * s + q) * MOL_HS + ht * 16 + r], xin[(4 * s + q) * MOL_XS + ft * 16 + r], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) dw1[(size_t)(ht * 16 + 4 * q + i) * 112 + ft * 16 + r] = acc[i];
        }
        // d h = d pre . W1  -> the other buffer
        for (int t = wave; t < ntr * 7; t += MOL_NW) {
            const int mt = t / 7, ft = t - mt * 7;
            v4 acc = v4{0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < HT; ++j) {
                const v4 av = *(const v4*)&pre[(mt * 16 + r) * MOL_HS + 16 * j + 4 * q];
                const v4 bv = *(const v4*)(m.p_w1pt + (size_t)(ft * 16 + r) * HP + 16 * j + 4 * q);
                acc = mfma4(av, bv, acc);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sim[(mt * 16 + 4 * q + i) * MOL_XS + ft * 16 + r] = acc[i];
        }
    }
    MOL_BAR();

    MOL_STAMP();   // readout backward
    // -------------------------------------------------------------------------------------- the layers, backward ----
    float* dh = sim;        // d loss / d (this layer's output h)
    float* oth = xin;       // the other atom-row buffer
    float* Cf = St;
    for (int li = a.nl - 1; li >= 0; --li) {
        MolLayerK& Y = a.layer[li];
        const bool last = li == a.nl - 1;
        const int FP = Y.FP, XS = FP + 4, KJ = Y.KJ;
        // (s0) every pair's g = d loss / d sim (+ chirality sign) and order -> coefficient tables (oth); score-weight partials
        float* gtab = oth;
        // (global loads of the phase first: the pair records of every degree's first round, this layer's input rows)
        const v4 pf1 = mol_pair_prefetch<1>(Y, m, tid), pf2 = mol_pair_prefetch<2>(Y, m, tid), pf3 = mol_pair_prefetch<3>(Y, m, tid),
                 pf4 = mol_pair_prefetch<4>(Y, m, tid);
        constexpr int XRN = (16 * NTR * 28 + MOL_THREADS - 1) / MOL_THREADS;      // 16-byte pieces of the input rows per thread
        v4 xrow[XRN];
#pragma unroll
        for (int u = 0; u < XRN; ++u) {
            const int it = tid + u * MOL_THREADS, j = it / (FP / 4), c4 = it - j * (FP / 4);
            xrow[u] = (it < NR * (FP / 4) && j < NA) ? *(const v4*)(Y.x_save + (size_t)(a0 + j) * FP + 4 * c4) : v4{0.f, 0.f, 0.f, 0.f};
        }
        mol_pairs_backward<1>(Y, last, dh, gtab, m, pf1, tid);
        mol_pairs_backward<2>(Y, last, dh, gtab, m, pf2, tid);
        mol_pairs_backward<3>(Y, last, dh, gtab, m, pf3, tid);
        mol_pairs_backward<4>(Y, last, dh, gtab, m, pf4, tid);
        MOL_BAR();
        MOL_STAMP();   // bwd layer: pairs
        if (tid < 12) {
            float s = 0.f;
            for (int w = 0; w < MOL_NW; ++w) s += m.red[w][tid];
            small[Y.theta_slot + tid] = s;
        }
        // (s1) d loss / d h is consumed: its buffer takes this layer's input rows (unit rows U; norms first)
        float* U = dh;
#pragma unroll
        for (int u = 0; u < XRN; ++u) {
            const int it = tid + u * MOL_THREADS, j = it / (FP / 4), c4 = it - j * (FP / 4);
            if (it < NR * (FP / 4)) *(v4*)&U[j * XS + 4 * c4] = xrow[u];
        }
        {
            float* eslab = slab + Y.slab_edge;
            mol_edge_grad_degree<1>(Y, gtab, bond, eslab, m, tid); mol_edge_grad_degree<2>(Y, gtab, bond, eslab, m, tid);
            mol_edge_grad_degree<3>(Y, gtab, bond, eslab, m, tid); mol_edge_grad_degree<4>(Y, gtab, bond, eslab, m, tid);
        }
        MOL_BAR();
        MOL_STAMP();   // bwd layer: input rows back, edge-support gradient
        mol_row_norms(U, NR, XS, FP, m, tid);
        MOL_BAR();
        for (int it = tid; it < NR * (FP / 4); it += MOL_THREADS) {
            const int j = it / (FP / 4), c4 = it - j * (FP / 4);
            v4 v = *(v4*)&U[j * XS + 4 * c4];
            const float iv = m.inv[j];
            v[0] *= iv; v[1] *= iv; v[2] *= iv; v[3] *= iv;
            *(v4*)&U[j * XS + 4 * c4] = v;
        }
        v4 gacc[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) gacc[mt] = v4{0.f, 0.f, 0.f, 0.f};
        const bool d4_valu = m.dcnt[3] <= MOL_D4_VALU_MAX;      // (chunk-uniform; the forward took the same branch)
        for (int p = 0; p < 2; ++p) {
            const int RT = Y.pass_rows[p] >> 4;
            if (p == 1 && d4_valu) { MOL_STAMP(); MOL_STAMP(); continue; }
            if (RT == 0) continue;
            // (the B operand of G += Cf . BankU for this pass: issued now, used behind the barrier)
            const int gft = wave & 7, gk0 = (wave >> 3) * MOL_GPW;       // this wave's feature tile and its first 16-row group
            v4 btv[MOL_GPW];
            {
                const float* bt = Y.bankUT + (size_t)((gft < KJ ? gft : 0) * 16 + r) * Y.RPT + Y.row_base[p] + 4 * q;
#pragma unroll
                for (int g = 0; g < MOL_GPW; ++g) btv[g] = (gk0 + g < RT && gft < KJ) ? *(const v4*)(bt + 16 * (gk0 + g)) : v4{0.f, 0.f, 0.f, 0.f};
            }
            // (s2) Cf [atoms, rows of the pass]: entry (j, (d, b, l)) = sum over j's neighbours n of degree d whose chosen order
            // maps the slot pointing at j to support b, of g(n, l) w_s / (W d); centre entries from the atom's own pairs
            if (p == 0) { mol_cf_degree<1>(Y, NR, gtab, Cf, m, tid); mol_cf_degree<2>(Y, NR, gtab, Cf, m, tid); mol_cf_degree<3>(Y, NR, gtab, Cf, m, tid); }
            else mol_cf_degree<4>(Y, NR, gtab, Cf, m, tid);
            for (int it = tid; it < NR * (Y.pass_rows[p] - Y.rows_real[p]); it += MOL_THREADS) {      // padding rows of the pass
                const int np = Y.pass_rows[p] - Y.rows_real[p];
                const int j = it / np, c = it - j * np;
                Cf[j * MOL_RS + Y.rows_real[p] + c] = 0.f;
            }
            MOL_BAR();
            MOL_STAMP();   // bwd layer: Cf of the pass
            // (s3) G += Cf . BankU  (wave = feature tile)
            if (gft < KJ) {
#pragma unroll
                for (int g = 0; g < MOL_GPW; ++g) {
                    if (gk0 + g >= RT) continue;
#pragma unroll
                    for (int mt = 0; mt < NT; ++mt) {
                        if (mt >= ntr) continue;
                        const v4 av = *(const v4*)&Cf[(mt * 16 + r) * MOL_RS + 16 * (gk0 + g) + 4 * q];
                        gacc[mt] = mfma4(av, btv[g], gacc[mt]);
                    }
                }
            }
            // (s4) Bg = Cf^T . U  -> this chunk's slab  (wave = bank-row tiles)
            {
                float* bslab = slab + Y.slab_bank + (size_t)Y.row_base[p] * FP;
                for (int rt = wave; rt < RT; rt += MOL_NW) {
                    float ar[4 * NT];
#pragma unroll
                    for (int s = 0; s < 4 * NT; ++s) ar[s] = s < ksa ? Cf[(4 * s + q) * MOL_RS + rt * 16 + r] : 0.f;
                    for (int ft = 0; ft < KJ; ++ft) {
                        v4 acc = v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 4 * NT; ++s)
                            if (s < ksa) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[s], U[(4 * s + q) * XS + ft * 16 + r], acc, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) bslab[(size_t)(rt * 16 + 4 * q + i) * FP + ft * 16 + r] = acc[i];
                    }
                }
            }
            MOL_BAR();
            MOL_STAMP();   // bwd layer: products of the pass
        }
        // (s5) G -> oth (the coefficient table is dead), then d loss / d (input rows) = inv (G - (G . u) u), in place
        if (d4_valu && Y.L[3] > 0) {
            // degree 4 on the vector pipe: part rows into the (now free) Cf region, bank rows straight into the slab (zeros
            // when the chunk has no such atom: the reduction sums every chunk's rows)
            if (m.dcnt[3] > 0) mol_bwd4_rows(Y, gtab, Cf, m, tid);
            mol_bwd4_bank(Y, gtab, U, slab + Y.slab_bank + (size_t)Y.row_base[1] * FP, m, tid);
            MOL_BAR();
        }
        if (li > 0 && tid < 16) m.mix[tid] = m.mixs[li - 1][tid];          // (this layer's are not read again)
        // (the contraction's parts in wave order: part 0 stores, the others add)
#pragma unroll
        for (int part = 0; part < MOL_KSPLIT; ++part) {
            if ((wave & 7) < KJ && (wave >> 3) == part) {
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (mt < ntr) {
                        float* dst = &oth[(mt * 16 + 4 * q + i) * XS + (wave & 7) * 16 + r];
                        *dst = part == 0 ? gacc[mt][i] : *dst + gacc[mt][i];
                    }
            }
            MOL_BAR();
        }
        if (d4_valu && Y.L[3] > 0 && m.dcnt[3] > 0) {        // G += the degree-4 part rows, in part order
            const int KV = FP >> 2, ne = m.dcnt[3] * 5;
            for (int it = tid; it < NR * KV; it += MOL_THREADS) {
                const int j = it / KV, k = it - j * KV;
                v4 v = *(v4*)&oth[j * XS + 4 * k];
                for (int e = 0; e < ne; ++e) if (m.tgt[e] == j) v += *(const v4*)&Cf[e * XS + 4 * k];
                *(v4*)&oth[j * XS + 4 * k] = v;
            }
            MOL_BAR();
        }
        for (int j0 = 0; j0 < NR; j0 += MOL_THREADS / 8) {
            const int j = j0 + (tid >> 3), s8 = tid & 7;
            float s = 0.f;
            if (j < NR) for (int f = 4 * s8; f < FP; f += 32) {
                const v4 gv = *(const v4*)&oth[j * XS + f], uv = *(const v4*)&U[j * XS + f];
                s = fmaf(gv[0], uv[0], s); s = fmaf(gv[1], uv[1], s); s = fmaf(gv[2], uv[2], s); s = fmaf(gv[3], uv[3], s);
            }
            s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            if (j < NR) {
                const float iv = m.inv[j], pr = m.big[j] ? s : 0.f;
                for (int f = 4 * s8; f < FP; f += 32) {
                    v4 gv = *(const v4*)&oth[j * XS + f];
                    const v4 uv = *(const v4*)&U[j * XS + f];
#pragma unroll
                    for (int c = 0; c < 4; ++c) gv[c] = j < NA ? iv * (gv[c] - pr * uv[c]) : 0.f;
                    *(v4*)&oth[j * XS + f] = gv;
                }
            }
        }
        MOL_BAR();
        MOL_STAMP();   // bwd layer: projection
        float* t_ = dh; dh = oth; oth = t_;
    }
    // dh = d loss / d (batch-norm output) [NAP, XS0]: d weight = sum dy * xhat, d bias = sum dy (this chunk's partial)
    {
        const int s8 = tid & 7, cw = tid >> 3, c = cw & 31, which = cw >> 5;      // 8 threads per (column, which)
        float s = 0.f;
        if (c < C) for (int j = s8; j < NA; j += 8) {
            const float dy = dh[j * XS0 + c];
            s += which ? dy : dy * ((a.x[(a0 + j) * a.xs + c] - m.bn_mu[c]) * m.bn_is[c]);
        }
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (c < C && s8 == 0) small[a.s_bn + which * 32 + c] = s;
    }
    MOL_STAMP();
#undef MOL_STAMP
}

template <int NT>
__global__ void __launch_bounds__(MOL_THREADS) molecule_step_kernel(MolArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    MolArgsP ap = (MolArgsP)__builtin_amdgcn_kernarg_segment_ptr();        // (`a` is the kernel's first and only argument)
    if constexpr (NT > 2) {
        const int m0 = a.chunk_ptr[blockIdx.x], m1 = a.chunk_ptr[blockIdx.x + 1];
        const int64_t na = a.mol_ptr[m1] - a.mol_ptr[m0];
        if (na > 48) { molecule_step_body<NT, NT>(ap, lds); return; }
        if (na > 32) { molecule_step_body<NT, 3>(ap, lds); return; }      // (molecules of 33 .. 48 atoms: three atom tiles, not four)
    }
    molecule_step_body<NT, 2>(ap, lds);
}

// ------------------------------------------------------------------------------------------------ reduction ----
struct MolReduceLayer {
    mkgnn_kernel_bank bank[4]; mkgnn_kernel_bank_grad grad[4];
    int F, FP, RPT, ER, E; int L[4];
    int sup_row[4], cen_row[4], row_base[2], e_row[4];
    size_t slab_bank, slab_edge; int theta_slot;
    int task0;                       // tasks: RPT bank rows, ER edge rows
};
struct MolReduceArgs {
    int nl; MolReduceLayer layer[MKGNN_MOLECULE_MAX_LAYERS];
    int task_w1, task_small, task_end;
    const float* slab; size_t slab_floats; int n_chunks;
    int H, G, HP, K3, C; size_t s_dw1; int s_loss, s_ffn, s_lin2, s_lin1b, s_bn;
    float *g_w1, *g_b1, *g_w2, *g_b2, *g_ffn_w, *g_ffn_b, *g_bn_w, *g_bn_b;
    float* loss; int n_mols; int mode;
    int64_t* rng; int64_t* rng_used; float head_drop;
};

// sum of element `e` (offset inside a chunk's slab) over the chunks [c0, c1), ascending, eight loads in flight
__device__ __forceinline__ float mol_sum_range(const float* slab, size_t stride, int c0, int c1, size_t e) {
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = c0; c < c1; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(c + u < c1 ? c + u : c) * stride + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (c + u < c1) t[u] += v[u];
    }
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}
__device__ __forceinline__ float mol_sum_chunks(const float* slab, size_t stride, int n, size_t e) {
    return mol_sum_range(slab, stride, 0, n, e);
}
// the same sum by the four waves of a block, a quarter of the chunks each, combined in wave order (sh: [4][128])
__device__ __forceinline__ void mol_sum_block(const float* slab, size_t stride, int n, size_t e0, int width, float* sh, float& g0, float& g1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = (n + 3) / 4, c0 = per * wv < n ? per * wv : n, c1 = c0 + per < n ? c0 + per : n;
    sh[wv * 128 + lane] = lane < width ? mol_sum_range(slab, stride, c0, c1, e0 + lane) : 0.f;
    sh[wv * 128 + 64 + lane] = lane + 64 < width ? mol_sum_range(slab, stride, c0, c1, e0 + lane + 64) : 0.f;
    __syncthreads();
    g0 = (sh[lane] + sh[128 + lane]) + (sh[256 + lane] + sh[384 + lane]);
    g1 = (sh[64 + lane] + sh[128 + 64 + lane]) + (sh[256 + 64 + lane] + sh[384 + 64 + lane]);
}

// One BLOCK per task (a bank / edge / lin1 row, or 64 of the small elements): its four waves each sum a quarter of the chunks'
// partials, ascending, and the quarters are combined in wave order -- fixed order, no float atomics.
__global__ void __launch_bounds__(256) molecule_reduce_kernel(MolReduceArgs a) {
    __shared__ float sh[512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int task = blockIdx.x;
    if (task >= a.task_end) return;
    if (task >= a.task_small) {
        // the small gradients, one element per lane; loss; the head's generator moves on
        const int e = (task - a.task_small) * 64 + lane;
        const int n_ffn = a.G, n_lin2 = a.G * a.H + a.G, n_b1 = a.H, n_bn = 64, n_theta = a.nl * 12;
        const bool head = (a.mode & MKGNN_MOLECULE_HEAD) != 0, bwd = (a.mode & MKGNN_MOLECULE_BACKWARD) != 0;
        long long off = -1; float* dst = nullptr; float scale = 1.f;
        int k = e;
        if (k == 0) { if (head && a.loss) { off = a.s_loss; dst = a.loss; scale = 1.f / (float)a.n_mols; } }
        else if (!bwd) {}
        else if (k == 1) { if (head && a.g_ffn_b) { off = a.s_loss + 1; dst = a.g_ffn_b; } }
        else if ((k -= 2) < n_ffn) { if (head && a.g_ffn_w) { off = a.s_ffn + k; dst = a.g_ffn_w + k; } }
        else if ((k -= n_ffn) < n_lin2) {
            off = a.s_lin2 + k;
            dst = k < a.G * a.H ? (a.g_w2 ? a.g_w2 + k : nullptr) : (a.g_b2 ? a.g_b2 + (k - a.G * a.H) : nullptr);
        }
        else if ((k -= n_lin2) < n_b1) { off = a.s_lin1b + k; dst = a.g_b1 ? a.g_b1 + k : nullptr; }
        else if ((k -= n_b1) < n_bn) {
            const int which = k >> 5, c = k & 31;
            if (c < a.C) { off = a.s_bn + k; dst = which == 0 ? (a.g_bn_w ? a.g_bn_w + c : nullptr) : (a.g_bn_b ? a.g_bn_b + c : nullptr); }
        }
        else if ((k -= n_bn) < n_theta) {
            const int li = k / 12, i = (k % 12) / 3, w = k % 3;
            const MolReduceLayer& Y = a.layer[li];
            if (Y.L[i] > 0) {
                off = Y.theta_slot + (k % 12);
                dst = w == 0 ? Y.grad[i].support_attr_sc_weight : (w == 1 ? Y.grad[i].center_attr_sc_weight : Y.grad[i].edge_attr_support_sc_weight);
            }
        }
        if (!dst) off = -1;
        const int per = (a.n_chunks + 3) / 4, c0 = per * wv < a.n_chunks ? per * wv : a.n_chunks, c1 = c0 + per < a.n_chunks ? c0 + per : a.n_chunks;
        sh[wv * 64 + lane] = off >= 0 ? mol_sum_range(a.slab, a.slab_floats, c0, c1, (size_t)off) : 0.f;
        __syncthreads();
        if (wv == 0 && off >= 0) dst[0] = ((sh[lane] + sh[64 + lane]) + (sh[128 + lane] + sh[192 + lane])) * scale;
        if (e == 0 && wv == 0 && head && a.head_drop > 0.f && a.rng) {
            const int64_t seed = a.rng[0], offset = a.rng[1];
            if (a.rng_used) { a.rng_used[0] = seed; a.rng_used[1] = offset; }
            a.rng[1] = offset + 1;
        }
        return;
    }
    if (!(a.mode & MKGNN_MOLECULE_BACKWARD)) return;
    if (task >= a.task_w1) {
        const int h = task - a.task_w1;                 // lin1 row h
        if (h >= a.H || !a.g_w1) return;
        float g0, g1;
        mol_sum_block(a.slab, a.slab_floats, a.n_chunks, a.s_dw1 + (size_t)h * 112, a.K3, sh, g0, g1);
        if (wv == 0) {
            if (lane < a.K3) a.g_w1[(size_t)h * a.K3 + lane] = g0;
            if (lane + 64 < a.K3) a.g_w1[(size_t)h * a.K3 + lane + 64] = g1;
        }
        return;
    }
    int li = 0;
    while (li + 1 < a.nl && task >= a.layer[li + 1].task0) ++li;
    const MolReduceLayer& Y = a.layer[li];
    int rrow = task - Y.task0;
    // bank row: sum the chunks' partial rows, then undo the unit normalisation against the raw row:
    // d s = (g - (g . s_hat) s_hat) / |s|   (|s| <= eps: g / eps)
    const float* src = nullptr; float* dst = nullptr; int width = 0; size_t e0 = 0;
    if (rrow < Y.RPT) {
        const int p = rrow >= Y.row_base[1] ? 1 : 0, rr = rrow - Y.row_base[p];
        for (int i = (p ? 3 : 0); i < (p ? 4 : 3); ++i) {
            const int d = i + 1, L = Y.L[i];
            if (rr >= Y.sup_row[i] && rr < Y.sup_row[i] + d * L) {
                const int b = (rr - Y.sup_row[i]) / L, l = (rr - Y.sup_row[i]) - b * L;
                src = Y.bank[i].x_support + ((size_t)l * d + b) * Y.F;
                dst = Y.grad[i].x_support ? Y.grad[i].x_support + ((size_t)l * d + b) * Y.F : nullptr;
            } else if (rr >= Y.cen_row[i] && rr < Y.cen_row[i] + L) {
                src = Y.bank[i].x_center + (size_t)(rr - Y.cen_row[i]) * Y.F;
                dst = Y.grad[i].x_center ? Y.grad[i].x_center + (size_t)(rr - Y.cen_row[i]) * Y.F : nullptr;
            }
        }
        width = Y.F; e0 = Y.slab_bank + (size_t)rrow * Y.FP;
    } else {
        rrow -= Y.RPT;
        if (rrow >= Y.ER) return;
        int i = 3;
        while (i > 0 && rrow < Y.e_row[i]) --i;
        const int d = i + 1, L = Y.L[i];
        const int b = (rrow - Y.e_row[i]) / L, l = (rrow - Y.e_row[i]) - b * L;
        src = Y.bank[i].edge_attr_support + ((size_t)l * d + b) * Y.E;
        dst = Y.grad[i].edge_attr_support ? Y.grad[i].edge_attr_support + ((size_t)l * d + b) * Y.E : nullptr;
        width = Y.E; e0 = Y.slab_edge + (size_t)rrow * 8;
    }
    if (!src || !dst) return;                            // (block-uniform)
    float g0, g1;
    mol_sum_block(a.slab, a.slab_floats, a.n_chunks, e0, width, sh, g0, g1);
    if (wv != 0) return;
    const bool ok0 = lane < width, ok1 = lane + 64 < width;
    const float s0 = ok0 ? src[lane] : 0.f, s1 = ok1 ? src[lane + 64] : 0.f;
    float ss = fmaf(s0, s0, 0.f);
    ss = fmaf(s1, s1, ss);
    ss = wave_sum(ss);
    const float nr = sqrtf(ss), iv = 1.f / fmaxf(nr, MKGNN_EPS);
    float dt = fmaf(g0, s0 * iv, 0.f);
    dt = fmaf(g1, s1 * iv, dt);
    dt = wave_sum(dt);
    if (!(nr > MKGNN_EPS)) dt = 0.f;
    if (ok0) dst[lane] = iv * (g0 - dt * (s0 * iv));
    if (ok1) dst[lane + 64] = iv * (g1 - dt * (s1 * iv));
}

}  // namespace mkgnn

using namespace mkgnn;

// ================================================================================================ C ABI ====
namespace {

struct MolShape {
    int nl, E, F0, H, G, HP, K3;
    int F[4], FP[4], L[4][4], off[4][4], K[4];
    int sup_row[4][4], cen_row[4][4], pass_rows[4][2], rows_real[4][2], row_base[4][2], RPT[4], e_row[4][4], ER[4];
    // slab layout (floats)
    int s_loss, s_ffn, s_lin2, s_lin1b, s_bn, theta_slot[4], small_floats;
    size_t s_dw1, slab_bank[4], slab_edge[4], slab_floats;
};

bool mol_shape(const mkgnn_molecule_net* net, int x_dim, MolShape& s) {
    if (!net || net->num_layers < 1 || net->num_layers > MKGNN_MOLECULE_MAX_LAYERS || net->E < 1 || net->E > 8) return false;
    if (x_dim < 1 || x_dim > 32) return false;
    s.nl = net->num_layers; s.E = net->E; s.F0 = x_dim;
    int prevK = x_dim;
    for (int li = 0; li < s.nl; ++li) {
        const mkgnn_molecule_layer& y = net->layer[li];
        if (y.F != prevK) return false;
        s.F[li] = y.F;
        s.FP[li] = li == 0 ? 32 : 112;
        if (y.F > s.FP[li]) return false;
        int K = 0;
        for (int i = 0; i < 4; ++i) {
            const int L = y.bank[i].num_kernels;
            if (L < 0 || L > 64) return false;
            s.L[li][i] = L; s.off[li][i] = K; K += L;
        }
        if (K < 1 || K > 112) return false;
        s.K[li] = K;
        int row = 0;
        for (int i = 0; i < 3; ++i) { s.sup_row[li][i] = row; row += (i + 1) * s.L[li][i]; s.cen_row[li][i] = row; row += s.L[li][i]; }
        s.rows_real[li][0] = row; s.pass_rows[li][0] = (row + 15) / 16 * 16;
        s.sup_row[li][3] = 0; s.cen_row[li][3] = 4 * s.L[li][3];
        s.rows_real[li][1] = 5 * s.L[li][3]; s.pass_rows[li][1] = (5 * s.L[li][3] + 15) / 16 * 16;
        if (s.pass_rows[li][0] > 256 || s.pass_rows[li][1] > 256) return false;
        s.row_base[li][0] = 0; s.row_base[li][1] = s.pass_rows[li][0];
        s.RPT[li] = s.pass_rows[li][0] + s.pass_rows[li][1];
        int er = 0;
        for (int i = 0; i < 4; ++i) { s.e_row[li][i] = er; er += (i + 1) * s.L[li][i]; }
        s.ER[li] = er;
        prevK = K;
    }
    s.K3 = prevK;
    s.H = net->readout.H; s.G = net->readout.G;
    if (net->readout.F != s.K3 || s.H < 1 || s.H > 64 || s.G < 1 || s.G > 64) return false;
    s.HP = (s.H + 15) / 16 * 16;
    // small slab: [loss, d ffn bias | d ffn weight G | d lin2 weight G*H, bias G | d lin1 bias H | bn 64 | theta nl*12]
    int o = 0;
    s.s_loss = o; o += 2;
    s.s_ffn = o; o += s.G;
    s.s_lin2 = o; o += s.G * s.H + s.G;
    s.s_lin1b = o; o += s.H;
    s.s_bn = o; o += 64;
    for (int li = 0; li < s.nl; ++li) { s.theta_slot[li] = o; o += 12; }
    s.small_floats = (o + 63) / 64 * 64;
    size_t f = (size_t)s.small_floats;
    s.s_dw1 = f; f += (size_t)s.HP * 112;
    for (int li = 0; li < s.nl; ++li) {
        s.slab_bank[li] = f; f += (size_t)s.RPT[li] * s.FP[li];
        s.slab_edge[li] = f; f += (size_t)((s.ER[li] * 8 + 63) / 64 * 64);
    }
    s.slab_floats = f;
    return true;
}

struct MolWs {
    size_t bankU[4], bankUT[4], edgeU[4], chir[4], mix[4], x_save[4], w1p, w1pt, bn_part, slab, total;
};

MolWs mol_ws(const MolShape& s, int64_t n_atoms, int64_t n_chunks) {
    MolWs w;
    size_t off = 0;
    for (int li = 0; li < s.nl; ++li) {
        w.bankU[li] = off;  off = align_up(off + (size_t)s.RPT[li] * s.FP[li] * 4);
        w.bankUT[li] = off; off = align_up(off + (size_t)s.RPT[li] * s.FP[li] * 4);
        w.edgeU[li] = off;  off = align_up(off + (size_t)(s.ER[li] + 1) * 8 * 4);
        w.chir[li] = off;   off = align_up(off + (size_t)s.L[li][3] * 12 + 16);
        w.mix[li] = off;    off = align_up(off + 64);
        w.x_save[li] = off; off = align_up(off + (size_t)n_atoms * s.FP[li] * 4);
    }
    w.w1p = off;  off = align_up(off + (size_t)s.HP * 112 * 4);
    w.w1pt = off; off = align_up(off + (size_t)s.HP * 112 * 4);
    w.bn_part = off; off = align_up(off + (size_t)MOL_BN_BLOCKS * 2 * 32 * 4);
    w.slab = off; off = align_up(off + (size_t)n_chunks * s.slab_floats * 4);
    w.total = off;
    return w;
}

size_t mol_lds_bytes(int NT) {
    const int NAP = 16 * NT;
    return (size_t)(2 * NAP * MOL_XS + NAP * MOL_RS + NAP * 32) * 4 + ((sizeof(MolMeta) + 15) & ~(size_t)15);
}
constexpr size_t MOL_LDS_MAX = 160 * 1024;

}  // namespace

static unsigned long long* g_mol_stamps = nullptr;

extern "C" {

// diagnostics: a device buffer of 256 uint64 that chunk 0 of the next step kernels writes its phase stamps into (null: off)
int mkgnn_debug_molecule_stamps(void* device_buffer) { g_mol_stamps = (unsigned long long*)device_buffer; return 0; }

int mkgnn_molecule_supported(const mkgnn_molecule_net* net, int32_t x_dim) {
    MolShape s;
    return mol_shape(net, x_dim, s) ? 1 : 0;
}

size_t mkgnn_molecule_workspace_bytes(const mkgnn_molecule_net* net, int32_t x_dim, int64_t n_atoms, int64_t n_chunks) {
    MolShape s;
    if (!mol_shape(net, x_dim, s) || n_atoms < 1 || n_chunks < 1) return 0;
    return mol_ws(s, n_atoms, n_chunks).total;
}

int mkgnn_molecule_step(const mkgnn_molecule_net* net, const mkgnn_molecule_batch* batch, int32_t mode,
                        const float* target, const float* grad_emb, float* emb, float* pred, float* loss,
                        void* workspace, size_t workspace_bytes, void* stream) {
    const char* who = "mkgnn_molecule_step";
    MolShape s;
    if (!net || !batch) return api_fail("%s: null net / batch", who);
    if (batch->n_atoms < 1 || batch->n_mols < 1 || batch->n_chunks < 1 || batch->n_atoms >= ((int64_t)1 << 31) ||
        batch->n_mols >= ((int64_t)1 << 31) || batch->n_chunks > batch->n_mols)
        return api_fail("%s: bad batch sizes", who);
    if (!batch->x || !batch->chunk_mol_ptr || !batch->mol_atom_ptr || !batch->atom_degree || !batch->atom_rank)
        return api_fail("%s: null batch pointer", who);
    if (!mol_shape(net, (int32_t)0 + (int32_t)net->layer[0].F, s))
        return api_fail("%s: model shape outside the molecule-resident kernels (mkgnn_molecule_supported)", who);
    if (batch->x_stride < s.F0) return api_fail("%s: bad x stride", who);
    const bool head = (mode & MKGNN_MOLECULE_HEAD) != 0, bwd = (mode & MKGNN_MOLECULE_BACKWARD) != 0, ext = (mode & MKGNN_MOLECULE_GRAD_EMB) != 0;
    if (!emb) return api_fail("%s: emb is null", who);
    if (head && (!target || !pred || !loss || !net->ffn_weight)) return api_fail("%s: HEAD needs target, pred, loss and the ffn weight", who);
    if (head && ext) return api_fail("%s: HEAD and GRAD_EMB exclude each other", who);
    if (bwd && !head && (!ext || !grad_emb)) return api_fail("%s: BACKWARD needs HEAD or GRAD_EMB with grad_emb", who);
    if (head && net->head_dropout > 0.f && !net->rng_state) return api_fail("%s: head dropout needs rng_state", who);
    if (!(net->head_dropout >= 0.f && net->head_dropout < 1.f)) return api_fail("%s: head dropout outside [0, 1)", who);
    if (!net->bn_training && (!net->bn_running_mean || !net->bn_running_var)) return api_fail("%s: eval-mode batch norm needs running statistics", who);
    for (int i = 0; i < 4; ++i) {
        const mkgnn_degree_bucket& b = batch->buckets[i];
        if (b.count > 0 && (!b.nei_index || !b.nei_edge_unit)) return api_fail("%s: degree %d bucket lacks nei_index / nei_edge_unit", who, i + 1);
    }
    for (int li = 0; li < s.nl; ++li)
        for (int i = 0; i < 4; ++i) {
            const mkgnn_kernel_bank& k = net->layer[li].bank[i];
            if (k.num_kernels > 0 && (!k.x_center || !k.x_support || !k.edge_attr_support || !k.support_attr_sc_weight ||
                                      !k.center_attr_sc_weight || !k.edge_attr_support_sc_weight))
                return api_fail("%s: layer %d degree %d bank has null parameters", who, li, i + 1);
            if (bwd && k.num_kernels > 0 && batch->buckets[i].count > 0 && !net->layer[li].saved[i].pair_state)
                return api_fail("%s: BACKWARD needs the pair records of layer %d degree %d", who, li, i + 1);
        }
    if (bwd && batch->buckets[3].count > 0 && s.L[s.nl - 1][3] > 0 && !net->layer[s.nl - 1].saved[3].chirality)
        return api_fail("%s: BACKWARD needs the chirality record of the last layer", who);
    const MolWs w = mol_ws(s, batch->n_atoms, batch->n_chunks);
    if (!workspace || workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return api_fail("%s: workspace too small or misaligned (%zu < %zu)", who, workspace_bytes, w.total);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;

    // ---- preparation: unit rows of every bank, mixing weights, chirality tables, lin1 padded; partial batch-norm statistics
    PrepArgsMol pa{};
    pa.nl = s.nl;
    int task = 0;
    for (int li = 0; li < s.nl; ++li) {
        PrepLayer& P = pa.layer[li];
        for (int i = 0; i < 4; ++i) { P.bank[i] = net->layer[li].bank[i]; P.L[i] = s.L[li][i]; P.sup_row[i] = s.sup_row[li][i]; P.cen_row[i] = s.cen_row[li][i]; P.e_row[i] = s.e_row[li][i]; }
        P.F = s.F[li]; P.FP = s.FP[li]; P.RPT = s.RPT[li]; P.ER = s.ER[li]; P.E = s.E;
        P.pass_rows[0] = s.pass_rows[li][0]; P.pass_rows[1] = s.pass_rows[li][1]; P.row_base[0] = 0; P.row_base[1] = s.row_base[li][1];
        P.bankU = (float*)(ws + w.bankU[li]); P.bankUT = (float*)(ws + w.bankUT[li]); P.edgeU = (float*)(ws + w.edgeU[li]);
        P.mix = (float*)(ws + w.mix[li]); P.chir = (int8_t*)(ws + w.chir[li]);
        P.task0 = task;
        task += s.RPT[li] + s.ER[li] + 1 + (s.L[li][3] * 12 + 63) / 64;
    }
    pa.task_w1 = task; task += s.HP; pa.task_end = task;
    pa.w1 = net->readout.lin1_weight; pa.H = s.H; pa.HP = s.HP; pa.K3 = s.K3;
    pa.w1p = (float*)(ws + w.w1p); pa.w1pt = (float*)(ws + w.w1pt);
    pa.task_blocks = (task + 3) / 4;
    pa.x = batch->x; pa.xs = batch->x_stride; pa.n = batch->n_atoms; pa.C = s.F0;
    pa.bn_part = (float*)(ws + w.bn_part);
    pa.bn_nblk = (int)((batch->n_atoms + 255) / 256 < MOL_BN_BLOCKS ? (batch->n_atoms + 255) / 256 : MOL_BN_BLOCKS);
    if (!net->readout.lin1_weight || !net->readout.lin2_weight) return api_fail("%s: readout weights are null", who);
    int prep_blocks = pa.task_blocks + (net->bn_training ? pa.bn_nblk : 0);
    pa.es_block = -1;
    if (net->edge_stats) {
        const mkgnn_bn_stats& e = *net->edge_stats;
        if (!e.x || e.n_rows < 1 || e.n_rows > MOL_EDGE_STATS_ROWS || e.C < 1 || e.C > 8 || e.x_stride < e.C || (e.row_key != nullptr) != (e.key_limit != nullptr))
            return api_fail("%s: edge_stats outside what the preparation launch takes (1..%lld rows, C <= 8, row_key and key_limit together): "
                            "use mkgnn_batchnorm_update_stats", who, (long long)MOL_EDGE_STATS_ROWS);
        pa.es = e;
        pa.es_block = prep_blocks++;
    }
    molecule_prepare_kernel<<<prep_blocks, 256, 0, st>>>(pa);

    // ---- the step
    MolArgs a{};
    a.n_atoms = batch->n_atoms; a.n_mols = (int)batch->n_mols; a.n_chunks = (int)batch->n_chunks;
    a.chunk_ptr = batch->chunk_mol_ptr; a.mol_ptr = batch->mol_atom_ptr; a.atom_deg = batch->atom_degree; a.atom_rank = batch->atom_rank;
    for (int i = 0; i < 4; ++i) { a.nei[i] = batch->buckets[i].nei_index; a.eunit[i] = batch->buckets[i].nei_edge_unit; }
    a.p_focal4 = batch->buckets[3].p_focal; a.nei_p4 = batch->buckets[3].nei_p;
    a.x = batch->x; a.xs = batch->x_stride; a.F0 = s.F0;
    a.bn_w = net->bn_weight; a.bn_b = net->bn_bias; a.run_mean = net->bn_running_mean; a.run_var = net->bn_running_var;
    a.nbt = net->bn_num_batches_tracked; a.bn_eps = net->bn_eps; a.bn_mom = net->bn_momentum; a.bn_training = net->bn_training;
    a.bn_part = pa.bn_part; a.bn_nblk = pa.bn_nblk;
    a.nl = s.nl;
    for (int li = 0; li < s.nl; ++li) {
        MolLayer& Y = a.layer[li];
        Y.F = s.F[li]; Y.FP = s.FP[li]; Y.KJ = s.FP[li] / 16; Y.K = s.K[li]; Y.RPT = s.RPT[li]; Y.ER = s.ER[li];
        for (int i = 0; i < 4; ++i) {
            Y.L[i] = s.L[li][i]; Y.off[i] = s.off[li][i]; Y.sup_row[i] = s.sup_row[li][i]; Y.cen_row[i] = s.cen_row[li][i]; Y.e_row[i] = s.e_row[li][i];
            Y.pair[i] = net->layer[li].saved[i].pair_state;
        }
        for (int p = 0; p < 2; ++p) { Y.pass_rows[p] = s.pass_rows[li][p]; Y.row_base[p] = s.row_base[li][p]; Y.rows_real[p] = s.rows_real[li][p]; }
        Y.bankU = pa.layer[li].bankU; Y.bankUT = pa.layer[li].bankUT; Y.edgeU = pa.layer[li].edgeU; Y.chir = pa.layer[li].chir; Y.mix = pa.layer[li].mix;
        Y.chir_out = li == s.nl - 1 ? net->layer[li].saved[3].chirality : nullptr;
        Y.x_save = (float*)(ws + w.x_save[li]);
        Y.sim_out = net->layer[li].sim_out; Y.sim_stride = net->layer[li].sim_stride;
        Y.slab_bank = s.slab_bank[li]; Y.slab_edge = s.slab_edge[li]; Y.theta_slot = s.theta_slot[li];
    }
    a.w1p = pa.w1p; a.w1pt = pa.w1pt; a.b1 = net->readout.lin1_bias; a.w2 = net->readout.lin2_weight; a.b2 = net->readout.lin2_bias;
    a.H = s.H; a.G = s.G; a.HP = s.HP; a.K3 = s.K3;
    a.mode = mode; a.ffn_w = net->ffn_weight; a.ffn_b = net->ffn_bias; a.y = target; a.head_drop = head ? net->head_dropout : 0.f; a.rng = net->rng_state;
    a.demb = grad_emb; a.emb = emb; a.pred = pred;
    a.stamps = g_mol_stamps;
    a.slab = (float*)(ws + w.slab); a.slab_floats = s.slab_floats;
    a.s_loss = s.s_loss; a.s_ffn = s.s_ffn; a.s_lin2 = s.s_lin2; a.s_lin1b = s.s_lin1b; a.s_bn = s.s_bn; a.s_dw1 = s.s_dw1;
    // chunks of at most 32 atoms take half the LDS and half the products
    {   // (the LDS ceiling is a per-device function attribute)
        static PerDeviceOnce attr_set;
        if (const int slot = attr_set.pending(); slot >= 0) {
            hipError_t e2 = hipFuncSetAttribute((const void*)molecule_step_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MOL_LDS_MAX);
            if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)molecule_step_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MOL_LDS_MAX);
            if (e2 != hipSuccess) return api_hip_fail(who, e2);
            attr_set.set(slot);
        }
    }
    if (batch->max_chunk_atoms < 1 || batch->max_chunk_atoms > MKGNN_MOLECULE_MAX_ATOMS)
        return api_fail("%s: max_chunk_atoms = %lld outside 1..%d", who, (long long)batch->max_chunk_atoms, MKGNN_MOLECULE_MAX_ATOMS);
    const bool small_chunks = batch->max_chunk_atoms <= 32;
    {   // the LDS table cache, if it fits behind this variant's buffers
        int o = 0;
        for (int li = 0; li < s.nl; ++li) { a.c_edge[li] = o; o += (s.ER[li] * 8 + 3) / 4 * 4; }
        a.c_w1p = o; o += s.HP * 112;
        a.c_w1pt = o; o += s.HP * 112;
        a.c_w2 = o; o += (s.G * s.H + 3) / 4 * 4;
        a.cache_floats = o;
        a.cache_on = mol_lds_bytes(small_chunks ? 2 : 4) + (size_t)o * 4 <= MOL_LDS_MAX ? 1 : 0;
    }
    const size_t lds_bytes = mol_lds_bytes(small_chunks ? 2 : 4) + (a.cache_on ? (size_t)a.cache_floats * 4 : 0);
    if (small_chunks) molecule_step_kernel<2><<<(unsigned)batch->n_chunks, MOL_THREADS, lds_bytes, st>>>(a);
    else molecule_step_kernel<4><<<(unsigned)batch->n_chunks, MOL_THREADS, lds_bytes, st>>>(a);

    // ---- reduction of the chunks' partials (and the loss; the head's generator moves on)
    MolReduceArgs ra{};
    ra.nl = s.nl;
    int rtask = 0;
    for (int li = 0; li < s.nl; ++li) {
        MolReduceLayer& Y = ra.layer[li];
        for (int i = 0; i < 4; ++i) { Y.bank[i] = net->layer[li].bank[i]; Y.grad[i] = net->layer[li].grad[i]; Y.L[i] = s.L[li][i];
                                      Y.sup_row[i] = s.sup_row[li][i]; Y.cen_row[i] = s.cen_row[li][i]; Y.e_row[i] = s.e_row[li][i]; }
        Y.F = s.F[li]; Y.FP = s.FP[li]; Y.RPT = s.RPT[li]; Y.ER = s.ER[li]; Y.E = s.E;
        Y.row_base[0] = 0; Y.row_base[1] = s.row_base[li][1];
        Y.slab_bank = s.slab_bank[li]; Y.slab_edge = s.slab_edge[li]; Y.theta_slot = s.theta_slot[li];
        Y.task0 = rtask; rtask += s.RPT[li] + s.ER[li];
    }
    ra.task_w1 = rtask; rtask += s.HP;
    ra.task_small = rtask; rtask += (2 + s.G + s.G * s.H + s.G + s.H + 64 + s.nl * 12 + 63) / 64;
    ra.task_end = rtask;
    ra.slab = a.slab; ra.slab_floats = s.slab_floats; ra.n_chunks = (int)batch->n_chunks;
    ra.H = s.H; ra.G = s.G; ra.HP = s.HP; ra.K3 = s.K3; ra.C = s.F0; ra.s_dw1 = s.s_dw1;
    ra.s_loss = s.s_loss; ra.s_ffn = s.s_ffn; ra.s_lin2 = s.s_lin2; ra.s_lin1b = s.s_lin1b; ra.s_bn = s.s_bn;
    ra.g_w1 = net->grad_lin1_weight; ra.g_b1 = net->grad_lin1_bias; ra.g_w2 = net->grad_lin2_weight; ra.g_b2 = net->grad_lin2_bias;
    ra.g_ffn_w = net->grad_ffn_weight; ra.g_ffn_b = net->grad_ffn_bias; ra.g_bn_w = net->grad_bn_weight; ra.g_bn_b = net->grad_bn_bias;
    ra.loss = loss; ra.n_mols = (int)batch->n_mols; ra.mode = mode & 7;
    ra.rng = net->rng_state; ra.rng_used = net->rng_used; ra.head_drop = head ? net->head_dropout : 0.f;
    if (head || bwd) molecule_reduce_kernel<<<rtask, 256, 0, st>>>(ra);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

}  // extern "C"
