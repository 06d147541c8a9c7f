// LDS-tiled backward kernels of the kernel convolution for the shapes the model uses (gfx950).
//
// The backward is sparse where the forward is dense: only the d entries of each d x d cosine
// matrix that the chosen permutation used carry a gradient, so both gradient products
//   rows:  g_xhat[n, slot, :]  = sum_l  coef[n, l] * unit_kernel_row[l, pi_{l,n}(slot), :]
//   bank:  g_unit[l, b, :]     = sum_n  coef[n, l] * xhat[n, pi^-1_{l,n}(b), :]
// are AXPYs with a data-dependent source row.  They run on the vector ALUs with the source rows
// in LDS (one 8-byte read per two FMAs, lanes across the feature axis, conflict-free), the
// per-(atom, kernel) coefficient and its 2-bit-packed permutation broadcast from LDS.
//
// Both kernels are persistent and software-pipelined: the atom ids of tile t+2 and the global
// loads (feature rows, output gradients, saved permutation ids) of tile t+1 are issued before
// tile t is multiplied, so the gather latency is hidden behind the LDS/FMA work.
//
// The bank product keeps its accumulators in registers across all tiles of a block and writes
// one partial slab per block; kc_backward_bank_reduce sums the slabs in a fixed order, so the
// result is reproducible bit for bit.
#include "kgnn_launch.h"

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int bank_li(int d) { return d == 1 ? 2 : (d == 2 ? 3 : (d == 3 ? 4 : 7)); }     // kernels per wave (8 waves)
__host__ __device__ constexpr int rows_cq(int d) { return d == 1 ? 2 : (d == 2 ? 3 : (d == 3 ? 4 : 7)); }    // coefficient pairs per thread
__host__ __device__ constexpr int bank_ta(int d) { return d <= 2 ? 32 : 16; }                                 // atoms per bank tile

// ------------------------------------------------------------------ rows ---
// Also accumulates the three score-weight partials (d sc / d theta_k = w_k (score_k - sc) / W).
template <int D, int KC>
__global__ void __launch_bounds__(256) kc_backward_rows_lds(BwdArgs a) {
    constexpr int FP = 16 * KC;
    constexpr int TA = 32;
    constexpr int CQ = rows_cq(D);
    extern __shared__ __align__(16) float lds[];
    __shared__ float red[3][4];
    const int L = a.L;
    float* bank = lds;                                              // [(D+1)*L][FP]
    float2* coef = (float2*)(lds + (size_t)(D + 1) * L * FP);       // [TA][L]  {g*ws/(W*D), packed pi}
    int* idbuf = (int*)(coef + (size_t)TA * L);                     // [2][TA] focal atom ids
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    copy_chunks_to_lds(bank, a.padded, (D + 1) * L * FP / 4, tid, [](int q) { return q; });
    const float w_s = a.mix[0], w_c = a.mix[1], w_e = a.mix[2], w_sum = a.mix[3];
    const float ws_n = w_s / w_sum / (float)D;
    const float ratio_c = w_c * (float)D / w_s;
    const int64_t ntiles = (a.n + TA - 1) / TA;
    const size_t ln = (size_t)L * a.n;
    const bool act = 2 * lane < FP;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;

    auto load_ids = [&](int64_t tile, int buf) {
        if (tid < TA) {
            int64_t n = tile * TA + tid;
            if (n >= a.n) n = a.n - 1;
            idbuf[buf * TA + tid] = (int)a.sel[n];
        }
    };
    float rg[CQ], rS[CQ], rC[CQ], rE[CQ];
    int ridx[CQ];
    auto fetch = [&](int64_t tile, int buf) {
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + 256 * k;
            const int i = q / L, l = q - i * L;
            const int64_t n = tile * TA + i;
            rg[k] = 0.f; ridx[k] = 0; rS[k] = rC[k] = rE[k] = 0.f;
            if (q < TA * L && n < a.n) {
                float g = a.gout[(int64_t)idbuf[buf * TA + i] * a.gs + a.off + l];
                if (a.chir) g *= (float)a.chir[(size_t)n * L + l];
                rg[k] = g;
                ridx[k] = a.best[(size_t)n * L + l];
                rS[k] = a.scores[(size_t)n * L + l];
                rC[k] = a.scores[ln + (size_t)n * L + l];
                rE[k] = a.scores[2 * ln + (size_t)n * L + l];
            }
        }
    };
    int64_t tile = blockIdx.x;
    int buf = 0;
    load_ids(tile, 0);
    __syncthreads();
    fetch(tile, 0);
    if (tile + gridDim.x < ntiles) load_ids(tile + gridDim.x, 1);
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + 256 * k;
            if (q < TA * L) {
                int pk = 0;
#pragma unroll
                for (int s = 0; s < D; ++s) pk |= perm_at<D>(ridx[k], s) << (2 * s);
                coef[q] = float2{rg[k] * ws_n, __int_as_float(pk)};
                const float sc = (rS[k] * w_s + rC[k] * w_c + rE[k] * w_e) / w_sum;
                p0 = fmaf(rg[k] * (w_s / w_sum), rS[k] - sc, p0);
                p1 = fmaf(rg[k] * (w_c / w_sum), rC[k] - sc, p1);
                p2 = fmaf(rg[k] * (w_e / w_sum), rE[k] - sc, p2);
            }
        }
        __syncthreads();
        const int64_t nxt = tile + gridDim.x;
        if (nxt < ntiles) fetch(nxt, buf ^ 1);
        // each wave owns 8 consecutive atoms of the tile and keeps 4 of them in flight: the chain
        // coefficient -> permutation bits -> row address -> row -> FMA is LDS-latency bound, four
        // independent chains hide it
        constexpr int AU = 4;
        for (int g4 = 0; g4 < TA / 4 / AU; ++g4) {
            const int i0 = wave * (TA / 4) + g4 * AU;
            if (tile * TA + i0 >= a.n) break;
            float2 acc[AU][D + 1];
#pragma unroll
            for (int u = 0; u < AU; ++u)
#pragma unroll
                for (int s = 0; s <= D; ++s) acc[u][s] = float2{0.f, 0.f};
            if (act) {
                const float* bl = bank + 2 * lane;
                for (int l = 0; l < L; ++l) {
                    const float2 vc = *(const float2*)(bl + (size_t)(D * L + l) * FP);
                    float2 c[AU];
#pragma unroll
                    for (int u = 0; u < AU; ++u) c[u] = coef[(size_t)(i0 + u) * L + l];    // zero beyond the bucket's end
#pragma unroll
                    for (int u = 0; u < AU; ++u) {
                        const int pk = __float_as_int(c[u].y);
                        const float gc = c[u].x * ratio_c;
                        acc[u][0].x = fmaf(gc, vc.x, acc[u][0].x);
                        acc[u][0].y = fmaf(gc, vc.y, acc[u][0].y);
#pragma unroll
                        for (int s = 0; s < D; ++s) {
                            const int b = (pk >> (2 * s)) & 3;
                            const float2 v = *(const float2*)(bl + (size_t)(b * L + l) * FP);
                            acc[u][1 + s].x = fmaf(c[u].x, v.x, acc[u][1 + s].x);
                            acc[u][1 + s].y = fmaf(c[u].x, v.y, acc[u][1 + s].y);
                        }
                    }
                }
            }
            if (2 * lane < a.F) {
#pragma unroll
                for (int u = 0; u < AU; ++u) {
                    const int64_t n = tile * TA + i0 + u;
                    if (n < a.n) {
#pragma unroll
                        for (int s = 0; s <= D; ++s)
                            *(float2*)(a.contrib + (size_t)(a.contrib_base + n * (D + 1) + s) * a.CS + 2 * lane) = acc[u][s];
                    }
                }
            }
        }
        __syncthreads();
        if (nxt + gridDim.x < ntiles) load_ids(nxt + gridDim.x, buf);
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
    if (lane == 0) { red[0][wave] = p0; red[1][wave] = p1; red[2][wave] = p2; }
    __syncthreads();
    if (tid < 3) a.theta_slab[(size_t)blockIdx.x * 4 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// ------------------------------------------------------------------ bank ---
template <int D, int KC, int LI, int TA>
__global__ void __launch_bounds__(512) kc_backward_bank_lds(BwdArgs a) {
    constexpr int NT = 512, NWV = 8;               // 8 waves share one staged tile: 2 waves per SIMD hide the LDS latency
    constexpr int FP = 16 * KC;
    constexpr int RS = FP + 8;                      // tile row: FP unit feature floats + 8 unit bond floats
    constexpr int CH = FP / 4;
    constexpr int NROW = TA * (D + 1);              // row = atom * (D+1) + slot, slot D = focal
    constexpr int MAXQ = (NROW * CH + NT - 1) / NT;
    constexpr int CQ = (TA * NWV * LI + NT - 1) / NT;   // coefficient pairs per thread (L <= NWV*LI)
    extern __shared__ __align__(16) float lds[];
    const int L = a.L;
    float* xt = lds;                                        // [NROW][RS]
    // per (atom, kernel): {g*ws/(W*D), byte offset of the tile row matched to support 0, 1, ..}; the offsets
    // are precomputed here so that the accumulate loop is add + ds_read + packed FMA per support
    constexpr int CW = (D == 1) ? 2 : ((D == 4) ? 8 : 4);   // floats per entry (D + 1 rounded up for aligned wide reads)
    // kernels padded to LP = NWV*LI per atom; the padding entries stay zero, so the accumulate loop needs no
    // bounds branch (a branch inside the unrolled loop would end each block with a full LDS wait)
    constexpr int LP = NWV * LI;
    float* coef = lds + NROW * RS;                          // [TA][LP][CW]
    int* idbuf = (int*)(coef + (size_t)TA * LP * CW);       // [2][NROW] atom ids
    float* invbuf = (float*)(idbuf + 2 * NROW);             // [2][NROW] 1/|x|
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float w_s = a.mix[0], w_c = a.mix[1], w_e = a.mix[2], w_sum = a.mix[3];
    const float ws_n = w_s / w_sum / (float)D;
    const float ratio_c = w_c * (float)D / w_s;
    const float ratio_e = w_e / w_s;
    f32x2 acc[LI][D + 1];                           // two features per lane, packed-FMA friendly
#pragma unroll
    for (int li = 0; li < LI; ++li)
#pragma unroll
        for (int s = 0; s <= D; ++s) acc[li][s] = f32x2{0.f, 0.f};
    const int64_t ntiles = (a.n + TA - 1) / TA;
    const bool feat = 2 * lane < FP;
    const bool act = 2 * lane < RS;

    auto load_ids = [&](int64_t tile, int buf) {
        for (int r = tid; r < NROW; r += NT) {
            const int i = r / (D + 1), slot = r - i * (D + 1);
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            const int64_t atom = (slot == D) ? a.sel[n] : a.nei[n * D + slot];
            idbuf[buf * NROW + r] = (int)atom;
            invbuf[buf * NROW + r] = a.inv[atom];
        }
    };
    f32x4 stage[MAXQ];
    float rg[CQ];
    int ridx[CQ];
    float ev[8];
    auto fetch = [&](int64_t tile, int buf) {
#pragma unroll
        // every load below is unconditional on a clamped address and masked afterwards: a load under a
        // lane-dependent branch ends its basic block with a full wait and serialises the round trips
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + NT * k;
            const int qc = q < NROW * CH ? q : NROW * CH - 1;
            const int row = qc / CH, c = qc - row * CH;
            const int cc = 4 * c < a.F ? c : 0;
            f32x4 v = *(const f32x4*)(a.x + (size_t)idbuf[buf * NROW + row] * a.xs + 4 * cc);
            if (4 * c >= a.F) v.x = 0.f;
            if (4 * c + 1 >= a.F) v.y = 0.f;
            if (4 * c + 2 >= a.F) v.z = 0.f;
            if (4 * c + 3 >= a.F) v.w = 0.f;
            stage[k] = v;
        }
        float rch[CQ];
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + NT * k;
            const int qc = q < TA * L ? q : TA * L - 1;
            const int i = qc / L, l = qc - i * L;
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            rg[k] = a.gout[(int64_t)idbuf[buf * NROW + i * (D + 1) + D] * a.gs + a.off + l];
            ridx[k] = a.best[(size_t)n * L + l];
            rch[k] = 1.f;
        }
        if (a.chir) {                                // one uniform branch for all the sign loads
#pragma unroll
            for (int k = 0; k < CQ; ++k) {
                const int q = tid + NT * k;
                const int qc = q < TA * L ? q : TA * L - 1;
                const int i = qc / L, l = qc - i * L;
                int64_t n = tile * TA + i;
                if (n >= a.n) n = a.n - 1;
                rch[k] = (float)a.chir[(size_t)n * L + l];
            }
        }
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + NT * k;
            const int i = q / L;
            const bool ok = q < TA * L && tile * TA + i < a.n;
            rg[k] = ok ? rg[k] * rch[k] : 0.f;
        }
        {                                            // bond vectors of (atom, slot); threads beyond TA*D re-read the last one
            const int t = tid < TA * D ? tid : TA * D - 1;
            const int i = t / D, slot = t - i * D;
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            const float* e = a.e_nei + (n * D + slot) * a.E;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float r = e[k < a.E ? k : a.E - 1];
                ev[k] = k < a.E ? r : 0.f;
            }
        }
    };
    for (int q = tid; q < TA * LP * CW; q += NT) coef[q] = 0.f;
    int64_t tile = blockIdx.x;
    int buf = 0;
    load_ids(tile, 0);
    __syncthreads();
    fetch(tile, 0);
    if (tile + gridDim.x < ntiles) load_ids(tile + gridDim.x, 1);
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        // ---- registers -> LDS: unit feature rows, unit bond vectors, coefficients
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + NT * k;
            if (q < NROW * CH) {
                const int row = q / CH, c = q - row * CH;
                *(f32x4*)(xt + (size_t)row * RS + 4 * c) = stage[k] * invbuf[buf * NROW + row];
            }
        }
        if (tid < TA * D) {
            const int i = tid / D, slot = tid - i * D;
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s2 = fmaf(ev[k], ev[k], s2);
            const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
            float* dst = xt + (size_t)(i * (D + 1) + slot) * RS + FP;
            *(f32x4*)dst = f32x4{ev[0] * ie, ev[1] * ie, ev[2] * ie, ev[3] * ie};
            *(f32x4*)(dst + 4) = f32x4{ev[4] * ie, ev[5] * ie, ev[6] * ie, ev[7] * ie};
        } else if (tid < TA * D + TA) {              // focal rows carry no bond vector
            float* dst = xt + (size_t)((tid - TA * D) * (D + 1) + D) * RS + FP;
            *(f32x4*)dst = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(dst + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + NT * k;
            if (q < TA * L) {
                float* ce = coef + ((size_t)(q / L) * LP + (q % L)) * CW;
                ce[0] = rg[k] * ws_n;
#pragma unroll
                for (int s = 0; s < D; ++s)          // support pi(s) takes the row of neighbour slot s
                    ce[1 + perm_at<D>(ridx[k], s)] = __int_as_float(s * RS * 4);
            }
        }
        __syncthreads();
        const int64_t nxt = tile + gridDim.x;
        if (nxt < ntiles) fetch(nxt, buf ^ 1);
        if (act) {
            const int64_t left = a.n - tile * TA;
            const int cnt = left < TA ? (int)left : TA;
            const float lane_mul = feat ? 1.f : ratio_e;       // bond columns carry w_e / w_s
#pragma unroll 1
            for (int i = 0; i < cnt; ++i) {
                const char* xr = (const char*)(xt + (size_t)i * (D + 1) * RS + 2 * lane);
                // phase 1: every coefficient entry of this atom (LI broadcast reads in flight together)
                float cv[LI][D + 1];
#pragma unroll
                for (int li = 0; li < LI; ++li) {
                    const float* ce = coef + ((size_t)i * LP + wave + NWV * li) * CW;
                    if constexpr (D == 1) { const f32x2 t = *(const f32x2*)ce; cv[li][0] = t.x; cv[li][1] = t.y; }
                    else if constexpr (D == 3) { const f32x4 t = *(const f32x4*)ce; cv[li][0] = t.x; cv[li][1] = t.y; cv[li][2] = t.z; cv[li][3] = t.w; }
                    else if constexpr (D == 2) { const f32x4 t = *(const f32x4*)ce; cv[li][0] = t.x; cv[li][1] = t.y; cv[li][2] = t.z; }
                    else { const f32x4 t = *(const f32x4*)ce; const float t4 = ce[4]; cv[li][0] = t.x; cv[li][1] = t.y; cv[li][2] = t.z; cv[li][3] = t.w; cv[li][4] = t4; }
                }
                const f32x2 vfocal = *(const f32x2*)(xr + D * RS * 4);
                // phase 2: all LI * D row reads, addresses straight from the entries
                f32x2 vv[LI][D];
#pragma unroll
                for (int li = 0; li < LI; ++li)
#pragma unroll
                    for (int b = 0; b < D; ++b) vv[li][b] = *(const f32x2*)(xr + __float_as_int(cv[li][1 + b]));
                // phase 3: packed FMAs
#pragma unroll
                for (int li = 0; li < LI; ++li) {
                    const float cc = cv[li][0] * lane_mul;
                    acc[li][D] += vfocal * (cv[li][0] * ratio_c);
#pragma unroll
                    for (int b = 0; b < D; ++b) acc[li][b] += vv[li][b] * cc;
                }
            }
        }
        __syncthreads();
        if (nxt + gridDim.x < ntiles) load_ids(nxt + gridDim.x, buf);
    }
    // ---- one partial slab per block (row order of kc_backward_bank in kgnn_generic.hip)
    float* slab = a.slab + (size_t)blockIdx.x * bank_floats(D, L, a.F, a.E);
    const size_t o_sup = (size_t)L * a.F, o_edg = o_sup + (size_t)L * D * a.F;
#pragma unroll
    for (int li = 0; li < LI; ++li) {
        const int l = wave + NWV * li;
        if (l < L) {
            if (2 * lane < a.F) {
                *(f32x2*)(slab + (size_t)l * a.F + 2 * lane) = acc[li][D];
#pragma unroll
                for (int b = 0; b < D; ++b)
                    *(f32x2*)(slab + o_sup + (size_t)(l * D + b) * a.F + 2 * lane) = acc[li][b];
            } else if (!feat && act) {
                const int e0 = 2 * lane - FP;
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    if (e0 < a.E) slab[o_edg + (size_t)(l * D + b) * a.E + e0] = acc[li][b].x;
                    if (e0 + 1 < a.E) slab[o_edg + (size_t)(l * D + b) * a.E + e0 + 1] = acc[li][b].y;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ host ---
static size_t rows_lds_bytes(int d, int FP, int L) { return ((size_t)(d + 1) * L * FP + 2 * 32 * (size_t)L + 2 * 32) * 4; }

bool lds_backward_supported(int d, int F, int E, int L, int64_t xs, const void* x) {
    if (d < 1 || d > 4 || L < 1 || E > 8 || (F & 1)) return false;
    const int FP = mfma_padded_width(F);
    if (!FP || xs % 4 != 0 || ((uintptr_t)x & 15)) return false;
    if (L > 8 * bank_li(d) || 32 * L > 256 * rows_cq(d)) return false;
    return rows_lds_bytes(d, FP, L) <= 160 * 1024 - 1024;
}

template <int D, int KC>
static hipError_t launch_lds_bwd(const BwdArgs& a0, int* nchunk_out, int* ntheta_out, bool rows_too, hipStream_t st) {
    constexpr int FP = 16 * KC;
    constexpr int LI = bank_li(D);
    constexpr int TA = bank_ta(D);
    BwdArgs a = a0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kc_backward_rows_lds<D, KC>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)kc_backward_bank_lds<D, KC, LI, TA>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (rows_too) {   // rows (also the score-weight partials)
        const size_t lds_bytes = rows_lds_bytes(D, FP, a.L);
        const int64_t ntiles = (a.n + 31) / 32;
        int per_cu = (int)((160 * 1024) / (lds_bytes + 256));
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 4) per_cu = 4;
        int64_t blocks = 256 * per_cu;
        if (blocks > THETA_SLAB_BLOCKS) blocks = THETA_SLAB_BLOCKS;
        if (blocks > ntiles) blocks = ntiles;
        kc_backward_rows_lds<D, KC><<<(int)blocks, 256, lds_bytes, st>>>(a);
        *ntheta_out = (int)blocks;
    }
    {   // bank
        constexpr int CW = (D == 1) ? 2 : ((D == 4) ? 8 : 4);
        const size_t lds_bytes = ((size_t)TA * (D + 1) * (FP + 8) + CW * TA * (size_t)(8 * LI) + 4 * TA * (D + 1)) * 4;
        const int64_t ntiles = (a.n + TA - 1) / TA;
        int64_t blocks = BWD_BANK_BLOCKS;
        if (blocks > ntiles) blocks = ntiles;
        a.nchunk = (int)blocks;
        kc_backward_bank_lds<D, KC, LI, TA><<<(int)blocks, 512, lds_bytes, st>>>(a);
        *nchunk_out = (int)blocks;
    }
    return hipGetLastError();
}

hipError_t launch_backward_lds(int d, const BwdArgs& a, int* nchunk_out, int* ntheta_out, bool rows_too, hipStream_t st) {
    const int KC = mfma_padded_width(a.F) / 16;
    if (KC == 2) {
        switch (d) {
            case 1: return launch_lds_bwd<1, 2>(a, nchunk_out, ntheta_out, rows_too, st);
            case 2: return launch_lds_bwd<2, 2>(a, nchunk_out, ntheta_out, rows_too, st);
            case 3: return launch_lds_bwd<3, 2>(a, nchunk_out, ntheta_out, rows_too, st);
            default: return launch_lds_bwd<4, 2>(a, nchunk_out, ntheta_out, rows_too, st);
        }
    }
    switch (d) {
        case 1: return launch_lds_bwd<1, 7>(a, nchunk_out, ntheta_out, rows_too, st);
        case 2: return launch_lds_bwd<2, 7>(a, nchunk_out, ntheta_out, rows_too, st);
        case 3: return launch_lds_bwd<3, 7>(a, nchunk_out, ntheta_out, rows_too, st);
        default: return launch_lds_bwd<4, 7>(a, nchunk_out, ntheta_out, rows_too, st);
    }
}

}  // namespace mkgnn
