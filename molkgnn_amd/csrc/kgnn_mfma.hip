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
// Block = 256 threads (4 waves), persistent over atom tiles.  LDS holds the whole unit-
// normalised kernel bank of this degree (loaded once per block) and one atom tile of RAW
// feature rows (raw because the chirality test compares raw rows bit for bit; 1/|x| is
// applied to the dot products).  The next tile's rows are fetched into registers while the
// current tile is being multiplied.  16-byte LDS chunks are XOR-swizzled so that the
// ds_read_b128 fragment reads are conflict-free (checked by brute force against the lane
// groups of MI355X_MICROARCH.md, LDS section).
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
    int nrt;        // row tiles (16 atoms) per block tile
    int nct;        // column tiles (<= 16 kernels)
    int kpt;        // kernels per column tile
};

// staged 16-byte chunks per thread the (D, KC) instantiation keeps in registers
__host__ __device__ constexpr int mfma_maxq(int D, int KC) { return (D == 4 && KC == 7) ? 9 : ((KC == 7) ? 14 : 4); }

__host__ __device__ static inline MfmaGeom mfma_geom(int L, int D, int KC) {
    MfmaGeom g;
    g.nct = (L + 15) / 16;
    g.kpt = (L + g.nct - 1) / g.nct;
    g.nrt = g.nct >= 3 ? 1 : 4 / g.nct;                       // at least four wave tiles per block tile
    const int fit = mfma_maxq(D, KC) * 256 / ((D + 1) * 16 * 4 * KC);   // row tiles the staging registers hold
    if (g.nrt > fit) g.nrt = fit < 1 ? 1 : fit;
    return g;
}

// LDS carve (floats).  Everything is a multiple of 4 floats.
struct MfmaLds {
    int bank, atile, ainv, esup, enei, idx, chirn, eqf, chirtab, total;
};

__host__ __device__ static inline MfmaLds mfma_lds(int D, int KC, int L, int nrt) {
    const int FP = 16 * KC, TA = 16 * nrt;
    MfmaLds o;
    int off = 0;
    o.bank = off;   off += (D + 1) * L * FP;
    o.atile = off;  off += (D + 1) * TA * FP;
    o.ainv = off;   off += (D + 1) * TA;
    o.esup = off;   off += D * L * 8;
    o.enei = off;   off += D * TA * 8;
    o.idx = off;    off += 2 * (D + 1) * TA;      // two tiles of atom ids (int32)
    o.chirn = off;  off += TA;
    o.eqf = off;    off += TA;
    o.chirtab = off; off += (L * 12 + 15) / 16 * 4;   // int8 table, rounded to 16 bytes
    o.total = off;
    return o;
}

template <int D, int KC>
__global__ void __launch_bounds__(256) kc_forward_mfma(FwdArgs a, MfmaGeom g) {
    constexpr int FP = 16 * KC;
    constexpr int CH = 4 * KC;                       // 16-byte chunks per row
    extern __shared__ __align__(16) float lds[];
    const int L = a.L;
    const int TA = 16 * g.nrt;
    const MfmaLds o = mfma_lds(D, KC, L, g.nrt);
    float* bank = lds + o.bank;
    float* atile = lds + o.atile;
    float* ainv = lds + o.ainv;
    float* esup = lds + o.esup;
    float* enei = lds + o.enei;
    int* idxbuf = (int*)(lds + o.idx);
    float* chirn = lds + o.chirn;
    int* eqf = (int*)(lds + o.eqf);
    int8_t* chirtab = (int8_t*)(lds + o.chirtab);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t ntiles = (a.n + TA - 1) / TA;
    const bool do_chir = (D == 4) && a.last;
    const int NROW = (D + 1) * TA;                   // staged rows per tile: slots 0..D-1 neighbours, slot D focal
    constexpr int MAXQ = mfma_maxq(D, KC);

    // ---- one-time: kernel bank -> LDS (swizzled by position inside its column tile)
    for (int q = tid; q < (D + 1) * L * CH; q += 256) {
        const int row = q / CH, c = q - row * CH;
        const int l = row % L;
        const f32x4 v = *(const f32x4*)(a.padded + (size_t)row * FP + 4 * c);
        *(f32x4*)(bank + (size_t)row * FP + 4 * (c ^ swz<KC>(l % g.kpt))) = v;
    }
    for (int q = tid; q < D * L * 2; q += 256)
        *(f32x4*)(esup + 4 * q) = *(const f32x4*)(a.edge_padded + 4 * q);
    if (do_chir)
        for (int q = tid; q < L * 12; q += 256) chirtab[q] = a.chir[q];
    if (tid < TA) eqf[tid] = 0;

    // atom ids of a tile -> LDS (clamped to the last valid atom; writes are masked later)
    auto load_ids = [&](int64_t tile, int buf) {
        if (tid < NROW) {
            const int slot = tid / TA, i = tid - slot * TA;
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            const int64_t id = (slot == D) ? a.sel[n] : a.nei[n * D + slot];
            idxbuf[buf * NROW + tid] = (int)id;
        }
    };
    f32x4 stage[MAXQ];
    // issue the global loads of a tile's feature rows into registers
    auto fetch_rows = [&](int buf) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + 256 * k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < NROW * CH) {
                const int row = q / CH, c = q - row * CH;
                if (4 * c < a.F) {
                    v = *(const f32x4*)(a.x + (size_t)idxbuf[buf * NROW + row] * a.xs + 4 * c);
                    if (4 * c + 4 > a.F) {            // zero what lies beyond the row's width
                        if (4 * c + 1 >= a.F) v.y = 0.f;
                        if (4 * c + 2 >= a.F) v.z = 0.f;
                        if (4 * c + 3 >= a.F) v.w = 0.f;
                    }
                }
            }
            stage[k] = v;
        }
    };
    auto store_rows = [&]() {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + 256 * k;
            if (q < NROW * CH) {
                const int row = q / CH, c = q - row * CH;
                *(f32x4*)(atile + (size_t)row * FP + 4 * (c ^ swz<KC>(row & 15))) = stage[k];
            }
        }
    };

    int64_t tile = blockIdx.x;
    int buf = 0;
    if (tile < ntiles) load_ids(tile, 0);
    __syncthreads();
    if (tile < ntiles) {
        fetch_rows(0);
        if (tile + gridDim.x < ntiles) load_ids(tile + gridDim.x, 1);
    }

    const float ws = a.mix[0], wc = a.mix[1], we = a.mix[2], wsum = a.mix[3];
    const int ci = lane & 15, kq = lane >> 4;

    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        // ---- stage this tile: rows, 1/|x|, unit bond vectors, chirality inputs
        store_rows();
        if (tid < NROW) ainv[tid] = a.inv[idxbuf[buf * NROW + tid]];
        for (int q = tid; q < D * TA; q += 256) {
            const int slot = q / TA, i = q - slot * TA;
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            const float* e = a.e_nei + (n * D + slot) * a.E;
            float v[8];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = k < a.E ? e[k] : 0.f; s = fmaf(v[k], v[k], s); }
            const float iv = 1.f / fmaxf(sqrtf(s), MKGNN_EPS);
            float* dst = enei + (size_t)q * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = v[k] * iv;
        }
        if constexpr (D == 4) {
            if (do_chir && tid < TA) {
                int64_t n = tile * TA + tid;
                if (n >= a.n) n = a.n - 1;
                float t[3][3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        t[j][c] = __fsub_rn(a.p_nei[(n * 4 + j) * 3 + c], a.p_focal[n * 3 + c]);
                chirn[tid] = triple_sign(t[0], t[1], t[2]);
            }
        }
        __syncthreads();
        // prefetch the next tile while this one is multiplied
        const int64_t nxt = tile + gridDim.x;
        if (nxt < ntiles) {
            fetch_rows(buf ^ 1);
        }
        if constexpr (D == 4) {
            // any two of the four neighbour rows bit-identical -> not chiral (kernels.py:310-317)
            if (do_chir) {
                for (int q = tid; q < TA * 6; q += 256) {
                    const int i = q / 6, pr = q - i * 6;
                    const int r0 = pr < 3 ? 0 : (pr < 5 ? 1 : 2);
                    const int r1 = pr < 3 ? pr + 1 : (pr < 5 ? pr - 1 : 3);
                    const float* ra = atile + (size_t)(r0 * TA + i) * FP;
                    const float* rb = atile + (size_t)(r1 * TA + i) * FP;
                    const int sw = swz<KC>(i & 15);
                    bool same = true;
                    for (int c = 0; c < CH && same; ++c) {
                        const f32x4 u = *(const f32x4*)(ra + 4 * (c ^ sw));
                        const f32x4 w = *(const f32x4*)(rb + 4 * (c ^ sw));
                        same = (u.x == w.x) && (u.y == w.y) && (u.z == w.z) && (u.w == w.w);
                    }
                    if (same) eqf[i] = 1;       // benign race: every writer stores 1
                }
                __syncthreads();                // flags are read by other waves' epilogues
            }
        }
        // ---- wave tiles: (row tile, column tile)
        for (int u = wave; u < g.nrt * g.nct; u += 4) {
            const int rt = u % g.nrt, ct = u / g.nrt;
            const int lcol = ct * g.kpt + ci;
            const bool col_ok = (ci < g.kpt) && (lcol < L);
            const int lrow = col_ok ? lcol : L - 1;
            const int swb = swz<KC>(lrow % g.kpt);
            const int swa = swz<KC>(ci);
            const float* arow = atile + (size_t)(rt * 16 + ci) * FP;
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
                f32x4 af[D + 1], bf[D + 1];
#pragma unroll
                for (int s = 0; s <= D; ++s) {
                    af[s] = *(const f32x4*)(arow + (size_t)s * TA * FP + 4 * (c ^ swa));
                    bf[s] = *(const f32x4*)(brow + (size_t)s * L * FP + 4 * (c ^ swb));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int b = 0; b < D; ++b)
                            cm[i][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][j], bf[b][j], cm[i][b], 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[D][j], bf[D][j], cc, 0, 0, 0);
                }
            }
            // ---- epilogue: lane = kernel lcol, atoms (lane >> 4) * 4 + jj
            int idx4[4];
            float best4[4], cen4[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int i = rt * 16 + kq * 4 + jj;          // atom inside the block tile
                float m[D][D];
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    const float iv = ainv[s * TA + i];
#pragma unroll
                    for (int b = 0; b < D; ++b) m[s][b] = cm[s][b][jj] * iv;
                }
                best_permutation<D>(m, best4[jj], idx4[jj]);
                cen4[jj] = cc[jj] * ainv[D * TA + i];
            }
            // edge matrices, one (a, b) tile at a time; keep the entry the chosen order uses
            float ed4[4][D];
            {
                const float* ea = enei + (size_t)(rt * 16 + ci) * 8 + 2 * kq;
                const float* eb = esup + (size_t)lrow * 8 + 2 * kq;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    const float2 av = *(const float2*)(ea + (size_t)s * TA * 8);
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        const float2 bv = *(const float2*)(eb + (size_t)b * L * 8);
                        f32x4 dm = {0.f, 0.f, 0.f, 0.f};
                        dm = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, dm, 0, 0, 0);
                        dm = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, dm, 0, 0, 0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
                            if (b == 0 || perm_at<D>(idx4[jj], s) == b) ed4[jj][s] = dm[jj];
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int i = rt * 16 + kq * 4 + jj;
                const int64_t n = tile * TA + i;
                float ed = ed4[jj][0];
#pragma unroll
                for (int s = 1; s < D; ++s) ed = __fadd_rn(ed, ed4[jj][s]);
                ed = ed / (float)D;
                float sc = __fadd_rn(__fadd_rn(__fmul_rn(best4[jj], ws), __fmul_rn(cen4[jj], wc)), __fmul_rn(ed, we)) / wsum;
                float ch = 1.f;
                if constexpr (D == 4) {
                    if (do_chir && !eqf[i]) ch = ((float)chirtab[lrow * 12 + idx4[jj]] == chirn[i]) ? 1.f : -1.f;
                    sc *= ch;
                }
                if (col_ok && n < a.n) {
                    const int64_t focal = idxbuf[buf * NROW + D * TA + i];
                    a.out[focal * a.os + a.off + lcol] = sc;
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
        // ---- the rest of every atom's row is zero (kernels.py:674-675)
        {
            const int rest = a.K - L;
            for (int q = tid; q < TA * rest; q += 256) {
                const int i = q / rest;
                int k = q - i * rest;
                if (k >= a.off) k += L;
                if (tile * TA + i < a.n) a.out[(int64_t)idxbuf[buf * NROW + D * TA + i] * a.os + k] = 0.f;
            }
        }
        __syncthreads();
        if (do_chir && tid < TA) eqf[tid] = 0;
        // ids for the tile after next go into the buffer this tile just released
        if (nxt + gridDim.x < ntiles) load_ids(nxt + gridDim.x, buf);
    }
}

// ---------------------------------------------------------------- host ----
bool mfma_forward_supported(int d, int F, int E, int L) {
    if (d < 1 || d > 4 || L < 1 || E > 8) return false;
    const int FP = mfma_padded_width(F);
    if (!FP) return false;
    const int KC = FP / 16;
    const MfmaGeom g = mfma_geom(L, d, KC);
    if (g.nrt * g.nct > 64) return false;
    // staged chunks per thread must fit the register array of the instantiation
    if (((d + 1) * 16 * g.nrt * 4 * KC + 255) / 256 > mfma_maxq(d, KC)) return false;
    return (size_t)mfma_lds(d, KC, L, g.nrt).total * 4 <= 160 * 1024;
}

template <int D, int KC>
static hipError_t launch_one(const FwdArgs& a, hipStream_t st) {
    const MfmaGeom g = mfma_geom(a.L, D, KC);
    const size_t lds_bytes = (size_t)mfma_lds(D, KC, a.L, g.nrt).total * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kc_forward_mfma<D, KC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int64_t ntiles = (a.n + 16 * g.nrt - 1) / (16 * g.nrt);
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int64_t blocks = 256 * per_cu;
    if (blocks > ntiles) blocks = ntiles;
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
