// MFMA forward of the kernel convolution (gfx950, exact-fp32 v_mfma_f32_16x16x4_f32).
//
// Why MFMA at all: per atom the N-hop layers do ~40 kflop against ~0.95 KB of compulsory
// traffic -- above the fp32 ridge, so the d x d cosine matrices are the cost,
// and they are a plain contraction [atoms x F] . [F x kernels] per (neighbour slot a, support
// slot b).  The f32-input MFMA is bit-for-bit an fmaf chain at the full fp32 rate with one
// operand register per lane, which no LDS-fed v_fma loop reaches.
//
// Tiling ("slot-major"): one MFMA tile = 16 atoms (rows) x 16 kernels (columns) for ONE
// (a, b) pair, so after the K loop every lane holds, for its kernel (lane & 15) and its four
// atoms ((lane >> 4) * 4 + j), the complete d x d matrix, the centre dot product and (later)
// the edge matrix IN REGISTERS: the permutation maximum needs no cross-lane traffic.
//
// Structure ("wave-autonomous"): the unit-normalised kernel bank of the degree sits in LDS,
// loaded once per block; after that there is NO block-level synchronisation.  Every wave walks
// its own 16-atom tiles: it gathers the A fragments of its atoms' feature rows (focal + d
// neighbours, CSR order) straight from global memory into registers -- lane (row i, k-quarter q)
// loads the 16-byte chunks 4t + q of row i, which is exactly the MFMA A-operand layout -- keeps
// them for all column tiles, and writes its atoms' full output rows.  Gather latency is hidden by
// the other waves of the CU (2-5 waves per SIMD for d <= 3; for d = 4 a tile carries ~60 k MFMA
// cycles, so the one-wave-per-SIMD exposure is a few percent) and by loading the next tile's atom
// ids while the current tile is multiplied.  Rows stay RAW (the chirality test compares raw rows
// bit for bit); 1/|x| multiplies the dot products.  The bank's 16-byte LDS chunks are
// XOR-swizzled so the ds_read_b128 fragment reads are conflict-free (brute-forced against the
// lane groups of MI355X_MICROARCH.md, LDS section).
#include "kgnn_launch.h"

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KC> __device__ __forceinline__ int swz(int pos) {
    if constexpr (KC == 7) {                // 112-float rows: rows 4 apart share banks
        return (0x1230 >> (((pos >> 2) & 3) * 4)) & 3;     // {0, 3, 2, 1}[(pos >> 2) & 3]
    } else {                                // 32-float rows: rows 2 apart share banks
        return (pos >> 1) & 7;
    }
}

struct MfmaGeom {
    int nct;        // column tiles (<= 16 kernels each)
    int kpt;        // kernels per column tile
    int cs;         // waves that share one atom tile, each taking every cs-th column tile
};

__host__ __device__ static inline MfmaGeom mfma_geom(int L) {
    MfmaGeom g;
    g.nct = (L + 15) / 16;
    g.kpt = (L + g.nct - 1) / g.nct;       // balanced: 10, 10, 15, 13 kernels for L = 10, 20, 30, 50
    g.cs = 1;
    return g;
}

// LDS carve (floats).  Everything is a multiple of 4 floats.
struct MfmaLds {
    int bank, esup, chirtab, total;
};

__host__ __device__ static inline MfmaLds mfma_lds(int D, int KC, int L) {
    const int FP = 16 * KC;
    MfmaLds o;
    int off = 0;
    o.bank = off;    off += (D + 1) * L * FP;
    o.esup = off;    off += D * L * 8;
    o.chirtab = off; off += (L * 12 + 15) / 16 * 4;   // int8 table, rounded to 16 bytes
    o.total = off;
    return o;
}

template <int D, int KC>
__global__ void __launch_bounds__(256, (D == 4 ? 1 : 2)) kc_forward_mfma(FwdArgs a, MfmaGeom g) {
    constexpr int FP = 16 * KC;
    constexpr int CH = 4 * KC;                       // 16-byte chunks per row
    constexpr int NW = 4;                            // waves per block
    extern __shared__ __align__(16) float lds[];
    const int L = a.L;
    const MfmaLds o = mfma_lds(D, KC, L);
    float* bank = lds + o.bank;
    float* esup = lds + o.esup;
    int8_t* chirtab = (int8_t*)(lds + o.chirtab);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool do_chir = (D == 4) && a.last;

    // ---- one-time: kernel bank -> LDS (swizzled by position inside its column tile)
    copy_chunks_to_lds(bank, a.padded, (D + 1) * L * CH, tid, [&](int q) {
        const int row = q / CH, c = q - row * CH;
        return row * CH + (c ^ swz<KC>((row % L) % g.kpt));
    });
    copy_chunks_to_lds(esup, a.edge_padded, D * L * 2, tid, [](int q) { return q; });
    if (do_chir)
        for (int q = tid; q < L * 12; q += 256) chirtab[q] = a.chir[q];
    __syncthreads();

    const float ws = a.mix[0], wc = a.mix[1], we = a.mix[2], wsum = a.mix[3];
    const int ci = lane & 15, kq = lane >> 4;
    const int64_t ntiles = (a.n + 15) / 16;
    // work unit = (atom tile, column part); units are dealt across blocks first so that a small
    // bucket still spreads over every CU
    const int64_t nunits = ntiles * g.cs;
    const int64_t ustride = (int64_t)gridDim.x * NW;
    int64_t unit = blockIdx.x + (int64_t)gridDim.x * wave;
    const uint32_t xs = (uint32_t)a.xs;

    // atom ids of this lane's row (slots 0..D-1 neighbours, slot D focal), one tile ahead
    uint32_t ids[D + 1];
    auto load_ids = [&](int64_t u) {
        int64_t n = (u / g.cs) * 16 + ci;
        if (n >= a.n) n = a.n - 1;
#pragma unroll
        for (int s = 0; s < D; ++s) ids[s] = (uint32_t)a.nei[n * D + s];
        ids[D] = (uint32_t)a.sel[n];
    };
    if (unit < nunits) load_ids(unit);

    for (; unit < nunits; unit += ustride) {
        const int64_t tile = unit / g.cs;
        const int cpart = (int)(unit - tile * g.cs);
        int64_t nrow = tile * 16 + ci;
        if (nrow >= a.n) nrow = a.n - 1;
        // ---- gather the A fragments: rows of (atom ci, slot s), chunks 4t + kq
        f32x4 af[D + 1][KC];
        float inv[D + 1];
#pragma unroll
        for (int s = 0; s <= D; ++s) {
            const float* row = a.x + (ids[s] * xs + 4u * kq);      // 32-bit offset (host checks N * stride < 2^32)
#pragma unroll
            for (int t = 0; t < KC - 1; ++t) af[s][t] = *(const f32x4*)(row + 16 * t);
            {   // only the last chunk can reach beyond the row's width (host guarantees FP - 16 < F <= FP)
                const int f0 = 16 * (KC - 1) + 4 * kq;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (f0 < a.F) {
                    v = *(const f32x4*)(row + 16 * (KC - 1));
                    if (f0 + 1 >= a.F) v.y = 0.f;
                    if (f0 + 2 >= a.F) v.z = 0.f;
                    if (f0 + 3 >= a.F) v.w = 0.f;
                }
                af[s][KC - 1] = v;
            }
            inv[s] = a.inv[ids[s]];
        }
        // unit bond vectors: lane holds components 2kq, 2kq+1 of (atom ci, slot s)
        float2 eu[D];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const float* e = a.e_nei + (nrow * D + s) * a.E;
            const float e0 = 2 * kq < a.E ? e[2 * kq] : 0.f;
            const float e1 = 2 * kq + 1 < a.E ? e[2 * kq + 1] : 0.f;
            float s2 = fmaf(e1, e1, e0 * e0);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
            eu[s] = float2{e0 * ie, e1 * ie};
        }
        // chirality, atom side (kernels.py:305-337)
        float sign_row = 0.f;
        int eq_row = 0;
        if constexpr (D == 4) {
            if (do_chir) {
                float t3[3][3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        t3[j][c] = __fsub_rn(a.p_nei[(nrow * 4 + j) * 3 + c], a.p_focal[nrow * 3 + c]);
                sign_row = triple_sign(t3[0], t3[1], t3[2]);
                // any two of the four neighbour rows bit-identical -> not chiral (:310-317)
                int diff = 0;     // bit k: pair k differs in this lane's chunks
                int k = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = i + 1; j < 4; ++j, ++k) {
                        bool same = true;
#pragma unroll
                        for (int t = 0; t < KC; ++t) {
                            const f32x4 u = af[i][t], w = af[j][t];
                            same = same && (u.x == w.x) && (u.y == w.y) && (u.z == w.z) && (u.w == w.w);
                        }
                        if (!same) diff |= 1 << k;
                    }
                diff |= __shfl_xor(diff, 16, 64);
                diff |= __shfl_xor(diff, 32, 64);
                eq_row = (diff != 0x3F);
            }
        }
        // next tile's ids travel while this tile is multiplied
        const uint32_t focal_row = ids[D];
        if (unit + ustride < nunits) load_ids(unit + ustride);
        // ---- values per accumulator row: atom kq*4 + jj of the tile
        float inv4[D + 1][4];
        uint32_t focal4[4];
        float sign4[4];
        int eq4[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int src = kq * 4 + jj;
#pragma unroll
            for (int s = 0; s <= D; ++s) inv4[s][jj] = __shfl(inv[s], src, 64);
            focal4[jj] = (uint32_t)__shfl((int)focal_row, src, 64);
            sign4[jj] = 0.f;
            eq4[jj] = 0;
            if constexpr (D == 4) {
                sign4[jj] = __shfl(sign_row, src, 64);
                eq4[jj] = __shfl(eq_row, src, 64);
            }
        }
        // ---- column tiles
        for (int ct = cpart; ct < g.nct; ct += g.cs) {
            const int lcol = ct * g.kpt + ci;
            const bool col_ok = (ci < g.kpt) && (lcol < L);
            const int lrow = col_ok ? lcol : L - 1;
            const int swb = swz<KC>(lrow % g.kpt);
            const float* brow = bank + (size_t)lrow * FP;
            f32x4 cm[D][D];
            f32x4 cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) cm[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                const int c = 4 * t + kq;
                f32x4 bf[D + 1];
#pragma unroll
                for (int s = 0; s <= D; ++s) bf[s] = *(const f32x4*)(brow + (size_t)s * L * FP + 4 * (c ^ swb));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int b = 0; b < D; ++b)
                            cm[i][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][t][j], bf[b][j], cm[i][b], 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[D][t][j], bf[D][j], cc, 0, 0, 0);
                }
            }
            // ---- epilogue: lane = kernel lcol, atoms kq*4 + jj
            int idx4[4];
            float best4[4], cen4[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float m[D][D];
#pragma unroll
                for (int s = 0; s < D; ++s)
#pragma unroll
                    for (int b = 0; b < D; ++b) m[s][b] = cm[s][b][jj] * inv4[s][jj];
                best_permutation<D>(m, best4[jj], idx4[jj]);
                cen4[jj] = cc[jj] * inv4[D][jj];
            }
            // bond-cosine matrices, one (a, b) tile at a time; keep the entry the chosen order uses
            float ed4[4][D];
            {
                const float* eb = esup + (size_t)lrow * 8 + 2 * kq;
#pragma unroll
                for (int s = 0; s < D; ++s) {
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        const float2 bv = *(const float2*)(eb + (size_t)b * L * 8);
                        f32x4 dm = {0.f, 0.f, 0.f, 0.f};
                        dm = __builtin_amdgcn_mfma_f32_16x16x4f32(eu[s].x, bv.x, dm, 0, 0, 0);
                        dm = __builtin_amdgcn_mfma_f32_16x16x4f32(eu[s].y, bv.y, dm, 0, 0, 0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            if (b == 0 || perm_at<D>(idx4[jj], s) == b) ed4[jj][s] = dm[jj];
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int64_t n = tile * 16 + kq * 4 + jj;
                float ed = ed4[jj][0];
#pragma unroll
                for (int s = 1; s < D; ++s) ed = __fadd_rn(ed, ed4[jj][s]);
                ed = div_by<D>(ed);
                float sc = __fadd_rn(__fadd_rn(__fmul_rn(best4[jj], ws), __fmul_rn(cen4[jj], wc)), __fmul_rn(ed, we)) / wsum;
                float ch = 1.f;
                if constexpr (D == 4) {
                    if (do_chir && !eq4[jj]) ch = ((float)chirtab[lrow * 12 + idx4[jj]] == sign4[jj]) ? 1.f : -1.f;
                    sc *= ch;
                }
                if (col_ok && n < a.n) {
                    a.out[(size_t)focal4[jj] * a.os + a.off + lcol] = sc;
                    if (a.best) a.best[(size_t)n * L + lcol] = (uint8_t)idx4[jj];
                    if (a.scores) {
                        const size_t ln = (size_t)L * a.n;
                        a.scores[(size_t)n * L + lcol] = best4[jj];
                        a.scores[ln + (size_t)n * L + lcol] = cen4[jj];
                        a.scores[2 * ln + (size_t)n * L + lcol] = ed;
                    }
                    if (a.chir_out) a.chir_out[(size_t)n * L + lcol] = (int8_t)ch;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- host ----
bool mfma_forward_supported(int d, int F, int E, int L) {
    if (d < 1 || d > 4 || L < 1 || E > 8) return false;
    const int FP = mfma_padded_width(F);
    if (!FP || F <= FP - 16) return false;          // only the last 16-float chunk may be partial
    if ((L + 15) / 16 > 64) return false;
    return (size_t)mfma_lds(d, FP / 16, L).total * 4 <= 160 * 1024;
}

template <int D, int KC>
static hipError_t launch_one(const FwdArgs& a, hipStream_t st) {
    MfmaGeom g = mfma_geom(a.L);
    const size_t lds_bytes = (size_t)mfma_lds(D, KC, a.L).total * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kc_forward_mfma<D, KC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int64_t ntiles = (a.n + 15) / 16;
    int per_cu = (int)((160 * 1024) / (lds_bytes + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 5) per_cu = 5;
    // few atom tiles: let the column tiles of one atom tile run on different waves
    if (ntiles < 768) g.cs = g.nct;
    int64_t blocks = 256 * per_cu;
    const int64_t need = (ntiles * g.cs + 3) / 4;
    if (blocks > need) blocks = need;
    kc_forward_mfma<D, KC><<<(int)blocks, 256, lds_bytes, st>>>(a, g);
    return hipGetLastError();
}

hipError_t launch_forward_mfma(int d, const FwdArgs& a, hipStream_t st) {
    if (a.n == 0 || a.L == 0) return hipSuccess;
    const int KC = mfma_padded_width(a.F) / 16;
    if (KC == 2) {
        switch (d) {
            case 1: return launch_one<1, 2>(a, st);
            case 2: return launch_one<2, 2>(a, st);
            case 3: return launch_one<3, 2>(a, st);
            default: return launch_one<4, 2>(a, st);
        }
    }
    switch (d) {
        case 1: return launch_one<1, 7>(a, st);
        case 2: return launch_one<2, 7>(a, st);
        case 3: return launch_one<3, 7>(a, st);
        default: return launch_one<4, 7>(a, st);
    }
}

}  // namespace mkgnn
