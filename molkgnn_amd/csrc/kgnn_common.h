// Shared device helpers and the workspace layout of the gfx950 kernel-convolution kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/molkgnn_hip.h"

#define MKGNN_EPS 1e-8f
#define MKGNN_WAVE 64

namespace mkgnn {

// Permutation tables of the reference (kernels.py:109-128): lexicographic for
// d <= 3, the 12 chirality-preserving orders for d = 4.  PERM[p][a] = pi_p(a).
__device__ __constant__ const int8_t PERM1[1][4] = {{0, 0, 0, 0}};
__device__ __constant__ const int8_t PERM2[2][4] = {{0, 1, 0, 0}, {1, 0, 0, 0}};
__device__ __constant__ const int8_t PERM3[6][4] = {{0, 1, 2, 0}, {0, 2, 1, 0}, {1, 0, 2, 0},
                                                    {1, 2, 0, 0}, {2, 0, 1, 0}, {2, 1, 0, 0}};
__device__ __constant__ const int8_t PERM4[12][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 0, 3, 2},
                                                     {1, 2, 0, 3}, {1, 3, 2, 0}, {2, 0, 1, 3}, {2, 1, 3, 0},
                                                     {2, 3, 0, 1}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 2, 1, 0}};

template <int D> struct PermInfo;
template <> struct PermInfo<1> { static constexpr int P = 1; };
template <> struct PermInfo<2> { static constexpr int P = 2; };
template <> struct PermInfo<3> { static constexpr int P = 6; };
template <> struct PermInfo<4> { static constexpr int P = 12; };

template <int D> __device__ __forceinline__ int perm_at(int p, int a) {
    if constexpr (D == 1) return 0;
    else if constexpr (D == 2) return PERM2[p][a];
    else if constexpr (D == 3) return PERM3[p][a];
    else return PERM4[p][a];
}

template <int D> struct PermC;
template <> struct PermC<1> { static constexpr int P = 1;  static constexpr int8_t t[1][4] = {{0, 0, 0, 0}}; };
template <> struct PermC<2> { static constexpr int P = 2;  static constexpr int8_t t[2][4] = {{0, 1, 0, 0}, {1, 0, 0, 0}}; };
template <> struct PermC<3> { static constexpr int P = 6;  static constexpr int8_t t[6][4] = {{0, 1, 2, 0}, {0, 2, 1, 0}, {1, 0, 2, 0}, {1, 2, 0, 0}, {2, 0, 1, 0}, {2, 1, 0, 0}}; };
template <> struct PermC<4> { static constexpr int P = 12; static constexpr int8_t t[12][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 3, 2, 0}, {2, 0, 1, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 2, 1, 0}}; };

// x / D, correctly rounded, without the hardware division sequence.  Powers of two are exact
// multiplies; for 3 the residual correction q + fma(-3, q, x) * (1/3) gives the IEEE quotient.
template <int D> __device__ __forceinline__ float div_by(float x) {
    if constexpr (D == 1) return x;
    else if constexpr (D == 2) return x * 0.5f;
    else if constexpr (D == 4) return x * 0.25f;
    else {
        const float c = 1.0f / 3.0f;
        const float q = x * c;
        const float r = fmaf(-3.0f, q, x);
        return fmaf(r, c, q);
    }
}

typedef float mkgnn_f32x4 __attribute__((ext_vector_type(4)));

// What the forward keeps of an (atom, kernel) pair for the backward (mkgnn_saved.pair_state): one 16-byte record
// {support score of the chosen permutation, centre score, edge score, index of the chosen permutation (int bits)} at
// pair o = n * L + l.  One store in the forward, one load in the backward (four scattered 4-byte accesses cost a wave
// several hundred cycles of issue time, measured in the streamed forward's epilogue).
__device__ __forceinline__ void pair_store(float* pair, size_t o, float S, float C, float Ed, int idx) {
    *(mkgnn_f32x4*)(pair + 4 * o) = mkgnn_f32x4{S, C, Ed, __int_as_float(idx)};
}
__device__ __forceinline__ mkgnn_f32x4 pair_load(const float* pair, size_t o) { return *(const mkgnn_f32x4*)(pair + 4 * o); }
__device__ __forceinline__ int pair_index(const float* pair, size_t o) { return __float_as_int(pair[4 * o + 3]); }
__device__ __forceinline__ int pair_index(mkgnn_f32x4 rec) { return __float_as_int(rec[3]); }

// Copy `n4` 16-byte chunks global -> LDS with `dst_of(q)` giving the destination chunk index;
// eight loads in flight per thread (a plain one-at-a-time loop serialises on the load latency).
template <typename DstOf>
__device__ __forceinline__ void copy_chunks_to_lds(float* lds_base, const float* src, int n4, int tid, DstOf dst_of) {
    for (int base = 0; base < n4; base += 256 * 8) {
        mkgnn_f32x4 tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = base + tid + 256 * k;
            if (q < n4) tmp[k] = *(const mkgnn_f32x4*)(src + 4 * (size_t)q);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = base + tid + 256 * k;
            if (q < n4) *(mkgnn_f32x4*)(lds_base + 4 * (size_t)dst_of(q)) = tmp[k];
        }
    }
}

// Best permutation of a d x d cosine matrix: every order scored as
// ((c0+c1)+c2)+c3 then / d, strict '>' scan in table order.
template <int D>
__device__ __forceinline__ void best_permutation(const float (&cm)[D][D], float& best, int& idx) {
    best = 0.f; idx = 0;
#pragma unroll
    for (int p = 0; p < PermC<D>::P; ++p) {
        float s = cm[0][PermC<D>::t[p][0]];
#pragma unroll
        for (int a = 1; a < D; ++a) s = __fadd_rn(s, cm[a][PermC<D>::t[p][a]]);
        s = div_by<D>(s);
        if (p == 0 || s > best) { best = s; idx = p; }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sign_f(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// sign(t3 . (t1 x t2)) with every product and sum rounded separately (no fma
// contraction), the op order of torch.cross followed by torch.dot
// (kernels.py:336-341).
__device__ __forceinline__ float triple_sign(const float* t1, const float* t2, const float* t3) {
    float cx = __fsub_rn(__fmul_rn(t1[1], t2[2]), __fmul_rn(t1[2], t2[1]));
    float cy = __fsub_rn(__fmul_rn(t1[2], t2[0]), __fmul_rn(t1[0], t2[2]));
    float cz = __fsub_rn(__fmul_rn(t1[0], t2[1]), __fmul_rn(t1[1], t2[0]));
    float dt = __fadd_rn(__fadd_rn(__fmul_rn(t3[0], cx), __fmul_rn(t3[1], cy)), __fmul_rn(t3[2], cz));
    return sign_f(dt);
}

// ---------------------------------------------------------------------------
// Workspace layout.  All offsets in bytes, 256-byte aligned.
// Per degree d the "prepared bank" holds the unit-normalised kernel rows and
// what the backward needs to undo the normalisation:
//   cen   [L, F]      x_center / max(|.|, eps)
//   sup   [L*d, F]    x_support rows
//   edg   [L*d, E]    edge_attr_support rows
//   icen  [L]  isup [L*d]  iedg [L*d]   1 / max(|row|, eps)
//   chir  [L, 12] int8  sign of the support tetrahedron for every order (d = 4)
//   mix   [4]  w_support, w_center, w_edge, their sum (kernels.py:402-422)
// ---------------------------------------------------------------------------
struct BankLayout {
    size_t cen, sup, edg, icen, isup, iedg, chir, mix;
    size_t padded;      // [(d+1)*L, FP] unit rows for the MFMA kernels: row b*L + l = support b of kernel l,
                        // row d*L + l = centre of kernel l; zero beyond F
    size_t edge_padded; // [d*L, 8] unit edge rows, row b*L + l, zero beyond E
    size_t end;
};

// Feature width the MFMA kernels pad to (0 = shape not covered by them).
__host__ __device__ static inline int mfma_padded_width(int F) { return F <= 32 ? 32 : (F <= 112 ? 112 : 0); }
// Row pitch of the padded bank copies (BankLayout::padded): round 1's kernels' width where they apply, whole 16-float
// chunks up to STREAM_MAX_F for the streamed kernels (KC = ceil(F / 16) <= 10 chunks: the (16, 32, 48, 64) banks' 160-wide
// N-hop rows), 0 = no padded copy.
constexpr int STREAM_MAX_F = 160;
__host__ __device__ static inline int bank_pitch(int F) {
    return F <= 112 ? mfma_padded_width(F) : (F <= STREAM_MAX_F ? (F + 15) / 16 * 16 : 0);
}

struct WorkspaceLayout {
    BankLayout bank[MKGNN_MAX_DEGREE];
    size_t eqflag;        // [N] int8: degree-4 atom has two identical neighbour rows (last layer)
    size_t signflag;      // [N] int8: sign of the degree-4 atom's neighbour tetrahedron (last layer)
    size_t contrib;       // [sum_d N_d (d+1), F] per-slot gradient rows (backward)
    size_t slab;          // partial bank gradients, per degree [nblk, bank floats]
    size_t slab_bytes_per_degree[MKGNN_MAX_DEGREE];
    size_t slab_off[MKGNN_MAX_DEGREE];
    size_t theta_off[MKGNN_MAX_DEGREE];
    size_t coefq_off[MKGNN_MAX_DEGREE];   // coefficient records of the streamed bank-gradient kernel, per degree
    size_t fwd_end;       // the forward needs [0, fwd_end)
    size_t total;
};

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

__host__ __device__ static inline size_t bank_floats(int d, int L, int F, int E) {
    // gradient rows of one bank: centre, supports, edge supports, 3 score weights (+1 pad)
    return (size_t)L * F + (size_t)L * d * F + (size_t)L * d * E + 4;
}

constexpr int BWD_BANK_BLOCKS = 256;   // persistent blocks of the LDS bank-gradient kernel
constexpr int SLAB_CHUNKS = 512;       // partial slabs per degree the workspace holds (streamed bank-gradient kernel: one per stream)
constexpr int THETA_SLAB_BLOCKS = 1024; // most blocks of the rows kernel (score-weight partials)

static inline WorkspaceLayout make_layout(const int32_t L[MKGNN_MAX_DEGREE], int F, int E,
                                          int64_t n_atoms, int64_t n_edges) {
    WorkspaceLayout w;
    size_t off = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        int d = i + 1;
        size_t l = (size_t)L[i];
        BankLayout& b = w.bank[i];
        b.cen = off;  off = align_up(off + l * F * 4);
        b.sup = off;  off = align_up(off + l * d * F * 4);
        b.edg = off;  off = align_up(off + l * d * E * 4);
        b.icen = off; off = align_up(off + l * 4);
        b.isup = off; off = align_up(off + l * d * 4);
        b.iedg = off; off = align_up(off + l * d * 4);
        b.chir = off; off = align_up(off + l * 12);
        b.mix = off;  off = align_up(off + 16);
        const int FP = bank_pitch(F);
        b.padded = off;      off = align_up(off + (size_t)(d + 1) * l * FP * 4);
        b.edge_padded = off; off = align_up(off + (size_t)d * l * 8 * 4);
        b.end = off;
    }
    w.eqflag = off;
    off = align_up(off + (size_t)n_atoms);
    w.signflag = off;
    off = align_up(off + (size_t)n_atoms);
    w.fwd_end = off;
    w.contrib = off;
    off = align_up(off + (size_t)(n_atoms + n_edges) * ((F + 3) / 4 * 4) * 4);
    w.slab = off;
    // an upper bound on the (atom tile, column tile) records of a degree whose bucket size is not known here
    const size_t max_tiles = (size_t)n_atoms / 16 + 4;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        w.slab_off[i] = off;
        w.slab_bytes_per_degree[i] = align_up(bank_floats(i + 1, L[i], F, E) * 4 * SLAB_CHUNKS);
        off += w.slab_bytes_per_degree[i];
        w.theta_off[i] = off;
        const size_t nct = L[i] > 0 ? (size_t)(L[i] + 15) / 16 : 0;
        const size_t theta_entries = max_tiles * nct > (size_t)THETA_SLAB_BLOCKS ? max_tiles * nct : (size_t)THETA_SLAB_BLOCKS;
        off += align_up(theta_entries * 4 * 4);
    }
    {   // the records of all degrees together hold at most max_tiles * max(nct) tiles' worth: sum_d N_d <= n_atoms
        size_t nct_max = 0;
        for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) { const size_t c = (size_t)(L[i] + 15) / 16; if (L[i] > 0 && c > nct_max) nct_max = c; }
        size_t per_degree_cap = (max_tiles + 4) * nct_max * 512 * 4;
        for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) w.coefq_off[i] = off;      // carved per call from the real bucket sizes
        off += align_up(per_degree_cap + 4 * 4096);
    }
    w.total = off;
    return w;
}

}  // namespace mkgnn
