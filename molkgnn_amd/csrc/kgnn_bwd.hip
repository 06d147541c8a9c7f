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
#include <cstring>

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
                const mkgnn_f32x4 rec = pair_load(a.pair, (size_t)n * L + l);
                rS[k] = rec[0]; rC[k] = rec[1]; rE[k] = rec[2];
                ridx[k] = pair_index(rec);
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

// Diagnostic cycle stamps of the bank kernel (tools/bwd_stamp_probe.py): per block, wave 0.
__device__ unsigned long long* g_bwd_stamp_buffer = nullptr;
// Compiled in only with -DMKGNN_BWD_STAMPS: the conditional stores perturb the wait-count placement.
#ifdef MKGNN_BWD_STAMPS
#define BWD_STAMP(slot)                                                                              \
    do {                                                                                             \
        if (stamps && tid == 0 && (slot) < 64) stamps[(slot)] = __builtin_readcyclecounter();        \
    } while (0)
#else
#define BWD_STAMP(slot) do { (void)stamps; (void)(slot); } while (0)
#endif

// ------------------------------------------------------------------ bank ---
// (device body: the block index and count are parameters, so that one launch can run the four degrees' blocks)
template <int D, int KC, int LI, int TA>
__device__ __forceinline__ void bank_body(const BwdArgs& a, const int vblock, const int vgrid) {
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
    // A tile row is RS / 2 lanes wide (60 for F = 110, 20 for F = 28).  An LDS read instruction costs the same
    // whether 20 or 64 lanes are active, and the accumulate loop is bound by exactly those instructions, so narrow
    // rows are packed: PACK atoms side by side in one wave (lane group g = lane / (RS / 2) takes atoms i + g),
    // each group accumulating its own partial sums, combined by shuffles in a fixed order at the end.
    constexpr int LPR = RS / 2;                      // lanes per row
    constexpr int PACK = 64 / LPR;                   // 1 (F = 110) or 3 (F = 28)
    const int grp = lane / LPR, ll = lane - grp * LPR;
    const bool feat = 2 * ll < FP;
    const bool act = grp < PACK;

    // Atom ids and row norms of a tile, one row per thread (NROW <= NT), pipelined over three tiles: ids are
    // loaded two tiles ahead, the norms (addressed by those ids) one tile ahead; both loads are issued BEFORE
    // the tile's other prefetch loads and the accumulate loop and land in LDS after it.  (Issued after the
    // loop they were exposed global round trips; issued after fetch() the in-order vmcnt wait for them
    // also waited for the whole next-tile row gather.)
    static_assert(NROW <= NT, "one id per thread");
    auto issue_inv = [&](int buf) -> float { return a.inv[idbuf[buf * NROW + (tid < NROW ? tid : NROW - 1)]]; };
    auto store_inv = [&](float v, int buf) {
        if (tid < NROW) invbuf[buf * NROW + tid] = v;
    };
    auto issue_id = [&](int64_t tile) -> int {
        const int r = tid < NROW ? tid : NROW - 1;
        const int i = r / (D + 1), slot = r - i * (D + 1);
        int64_t n = tile * TA + i;
        if (n >= a.n) n = a.n - 1;
        const int64_t* src = (slot == D) ? (a.sel + n) : (a.nei + n * D + slot);
        return (int)*src;
    };
    auto store_id = [&](int id, int buf) {
        if (tid < NROW) idbuf[buf * NROW + tid] = id;
    };
    f32x4 stage[MAXQ];
    float rg[CQ], rS[CQ], rC[CQ], rE[CQ];
    int ridx[CQ], rch[CQ];
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;              // score-weight partials (when the rows kernel leaves them to us)
    float ev;                                        // one bond component per thread: (atom, slot) = tid / 8, k = tid % 8
    static_assert(TA * D * 8 <= NT, "one bond component per thread");
    const int8_t* chp = a.chir ? a.chir : (const int8_t*)a.pair;      // always loadable; ignored when there are no signs
    // fetch() only ISSUES loads (unconditional, clamped addresses) and leaves the raw values in registers;
    // every use of a loaded value -- column masks, validity masks, products -- is in the staging code at the
    // top of the next iteration, and there is no branch in here: a use, or the end of a conditional block
    // that contains loads, makes the compiler wait for them on the spot -- a full global round trip per
    // tile, exposed.  Past the last tile the (clamped) loads are simply not used.
    auto fetch = [&](int64_t tile, int buf) {
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + NT * k;
            const int qc = q < NROW * CH ? q : NROW * CH - 1;
            const int row = qc / CH, c = qc - row * CH;
            const int cc = 4 * c < a.F ? c : 0;
            stage[k] = *(const f32x4*)(a.x + (size_t)idbuf[buf * NROW + row] * a.xs + 4 * cc);
        }
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + NT * k;
            const int qc = q < TA * L ? q : TA * L - 1;
            const int i = qc / L, l = qc - i * L;
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            rg[k] = a.gout[(int64_t)idbuf[buf * NROW + i * (D + 1) + D] * a.gs + a.off + l];
            const mkgnn_f32x4 rec = pair_load(a.pair, (size_t)n * L + l);
            ridx[k] = pair_index(rec);
            rch[k] = chp[(size_t)n * L + l];
            // the three cosine scores feed d sc / d theta_k = w_k (score_k - sc) / W  (SURVEY 8 a-9); this kernel visits
            // every (atom, kernel) pair exactly once, with one thread per pair: the cheapest place to sum them
            rS[k] = rec[0]; rC[k] = rec[1]; rE[k] = rec[2];
        }
        {
            const int t = (tid >> 3) < TA * D ? (tid >> 3) : TA * D - 1, k = tid & 7;
            const int i = t / D, slot = t - i * D;
            int64_t n = tile * TA + i;
            if (n >= a.n) n = a.n - 1;
            ev = a.e_nei[(n * D + slot) * a.E + (k < a.E ? k : a.E - 1)];
        }
    };
    unsigned long long* stamps = g_bwd_stamp_buffer ? g_bwd_stamp_buffer + ((size_t)(D - 1) * 256 + vblock) * 64 : nullptr;
    int slot = 2;
    BWD_STAMP(0);
    for (int q = tid; q < TA * LP * CW; q += NT) coef[q] = 0.f;
    int64_t tile = vblock;
    int buf = 0;
    store_id(issue_id(tile), 0);
    store_id(issue_id(tile + vgrid), 1);         // (clamped past the end: harmless)
    __syncthreads();
    store_inv(issue_inv(0), 0);
    fetch(tile, 0);
    __syncthreads();                                 // the first staging reads other threads' norms
    BWD_STAMP(1);
    for (; tile < ntiles; tile += vgrid, buf ^= 1) {
        // ---- registers -> LDS: unit feature rows, unit bond vectors, coefficients
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = tid + NT * k;
            if (q < NROW * CH) {
                const int row = q / CH, c = q - row * CH;
                f32x4 v = stage[k];
                if (4 * c >= a.F) v.x = 0.f;
                if (4 * c + 1 >= a.F) v.y = 0.f;
                if (4 * c + 2 >= a.F) v.z = 0.f;
                if (4 * c + 3 >= a.F) v.w = 0.f;
                *(f32x4*)(xt + (size_t)row * RS + 4 * c) = v * invbuf[buf * NROW + row];
            }
        }
        {   // unit bond vectors: eight lanes per (atom, slot), norm by an xor tree over them
            const float e = (tid & 7) < a.E ? ev : 0.f;
            float s2 = e * e;
            s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
            const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
            if ((tid >> 3) < TA * D) {
                const int t = tid >> 3, i = t / D, slot = t - i * D;
                xt[(size_t)(i * (D + 1) + slot) * RS + FP + (tid & 7)] = e * ie;
            }
            if (tid < TA * 8) xt[(size_t)((tid >> 3) * (D + 1) + D) * RS + FP + (tid & 7)] = 0.f;   // focal rows carry no bond vector
        }
#pragma unroll
        for (int k = 0; k < CQ; ++k) {
            const int q = tid + NT * k;
            if (q < TA * L) {
                float* ce = coef + ((size_t)(q / L) * LP + (q % L)) * CW;
                const float sgn = a.chir ? (float)(int8_t)rch[k] : 1.f;
                const float g = (tile * TA + q / L < a.n) ? rg[k] * sgn : 0.f;        // atoms past the end contribute nothing
                ce[0] = g * ws_n;
                const float sc = (rS[k] * w_s + rC[k] * w_c + rE[k] * w_e) / w_sum;
                p0 = fmaf(g * (w_s / w_sum), rS[k] - sc, p0);
                p1 = fmaf(g * (w_c / w_sum), rC[k] - sc, p1);
                p2 = fmaf(g * (w_e / w_sum), rE[k] - sc, p2);
#pragma unroll
                for (int s = 0; s < D; ++s)          // support pi(s) takes the row of neighbour slot s
                    ce[1 + perm_at<D>(ridx[k], s)] = __int_as_float(s * RS * 4);
            }
        }
        BWD_STAMP(slot);
        __syncthreads();
        BWD_STAMP(slot + 1);
        const int64_t nxt = tile + vgrid;
        const int id_ahead = issue_id(nxt + vgrid);      // these two first: the waits for them must not
        const float inv_ahead = issue_inv(buf ^ 1);          // cover the row gather issued next
        fetch(nxt < ntiles ? nxt : tile, buf ^ 1);
        BWD_STAMP(slot + 2);
        if (act) {
            const int64_t left = a.n - tile * TA;
            const int cnt = left < TA ? (int)left : TA;
            const float lane_mul = feat ? 1.f : ratio_e;       // bond columns carry w_e / w_s
            auto read_coef = [&](int i, float (&cv)[LI][D + 1]) {
#pragma unroll
                for (int li = 0; li < LI; ++li) {
                    const float* ce = coef + ((size_t)i * LP + wave + NWV * li) * CW;
                    if constexpr (D == 1) { const f32x2 t = *(const f32x2*)ce; cv[li][0] = t.x; cv[li][1] = t.y; }
                    else if constexpr (D == 3) { const f32x4 t = *(const f32x4*)ce; cv[li][0] = t.x; cv[li][1] = t.y; cv[li][2] = t.z; cv[li][3] = t.w; }
                    else if constexpr (D == 2) { const f32x4 t = *(const f32x4*)ce; cv[li][0] = t.x; cv[li][1] = t.y; cv[li][2] = t.z; }
                    else { const f32x4 t = *(const f32x4*)ce; const float t4 = ce[4]; cv[li][0] = t.x; cv[li][1] = t.y; cv[li][2] = t.z; cv[li][3] = t.w; cv[li][4] = t4; }
                }
            };
            // The coefficient entries of atom i + 1 are read while atom i's rows are in flight, so an atom costs
            // one LDS round trip (its row reads) instead of two dependent ones.
            float cv[LI][D + 1];
            read_coef(grp < cnt ? grp : cnt - 1, cv);
#pragma unroll 1
            for (int i0 = 0; i0 < cnt; i0 += PACK) {
                const int i = i0 + grp < cnt ? i0 + grp : cnt - 1;        // this lane group's atom (clamped: masked below)
                if constexpr (PACK > 1) {
                    if (i0 + grp >= cnt) {
#pragma unroll
                        for (int li = 0; li < LI; ++li) cv[li][0] = 0.f;
                    }
                }
                const char* xr = (const char*)(xt + (size_t)i * (D + 1) * RS + 2 * ll);
                const f32x2 vfocal = *(const f32x2*)(xr + D * RS * 4);
                // all LI * D row reads, addresses straight from the entries
                f32x2 vv[LI][D];
#pragma unroll
                for (int li = 0; li < LI; ++li)
#pragma unroll
                    for (int b = 0; b < D; ++b) vv[li][b] = *(const f32x2*)(xr + __float_as_int(cv[li][1 + b]));
                float c0[LI];
#pragma unroll
                for (int li = 0; li < LI; ++li) c0[li] = cv[li][0];
                read_coef(i + PACK < cnt ? i + PACK : cnt - 1, cv);   // (re-read of a clamped atom is zeroed at the loop top)
                // packed FMAs
#pragma unroll
                for (int li = 0; li < LI; ++li) {
                    const float cc = c0[li] * lane_mul;
                    acc[li][D] += vfocal * (c0[li] * ratio_c);
#pragma unroll
                    for (int b = 0; b < D; ++b) acc[li][b] += vv[li][b] * cc;
                }
            }
        }
        BWD_STAMP(slot + 3);
        store_inv(inv_ahead, buf ^ 1);                       // read by the next tile's staging, after its barrier... and
        __syncthreads();                                     // ...this one orders it before that staging
        store_id(id_ahead, buf);
        BWD_STAMP(slot + 4);
        slot += 5;
    }
    BWD_STAMP(62);
    if (a.theta_in_bank) {                           // fixed-order block sum of the score-weight partials
        __syncthreads();
        float* red = coef;                           // the coefficient image is free now
        p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
        if (lane == 0) { red[wave] = p0; red[NWV + wave] = p1; red[2 * NWV + wave] = p2; }
        __syncthreads();
        if (tid < 3) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t += red[tid * NWV + w];
            a.theta_slab[(size_t)vblock * 4 + tid] = t;
        }
    }
    if constexpr (PACK > 1) {                        // lane groups -> group 0, fixed order (g0 + g1) + g2 ...
#pragma unroll
        for (int li = 0; li < LI; ++li)
#pragma unroll
            for (int s = 0; s <= D; ++s) {
                f32x2 t = acc[li][s];
#pragma unroll
                for (int g = 1; g < PACK; ++g) {
                    const int src = (lane + g * LPR) & 63;
                    t.x += __shfl(acc[li][s].x, src, 64);
                    t.y += __shfl(acc[li][s].y, src, 64);
                }
                acc[li][s] = t;
            }
    }
    const bool writer = grp == 0;
    // ---- one partial slab per block (row order of kc_backward_bank in kgnn_generic.hip)
    float* slab = a.slab + (size_t)vblock * bank_floats(D, L, a.F, a.E);
    const size_t o_sup = (size_t)L * a.F, o_edg = o_sup + (size_t)L * D * a.F;
#pragma unroll
    for (int li = 0; li < LI; ++li) {
        const int l = wave + NWV * li;
        if (l < L) {
            if (writer && 2 * ll < a.F) {
                *(f32x2*)(slab + (size_t)l * a.F + 2 * ll) = acc[li][D];
#pragma unroll
                for (int b = 0; b < D; ++b)
                    *(f32x2*)(slab + o_sup + (size_t)(l * D + b) * a.F + 2 * ll) = acc[li][b];
            } else if (writer && !feat) {
                const int e0 = 2 * ll - FP;
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    if (e0 < a.E) slab[o_edg + (size_t)(l * D + b) * a.E + e0] = acc[li][b].x;
                    if (e0 + 1 < a.E) slab[o_edg + (size_t)(l * D + b) * a.E + e0 + 1] = acc[li][b].y;
                }
            }
        }
    }
}

template <int D, int KC, int LI, int TA>
__global__ void __launch_bounds__(512) kc_backward_bank_lds(BwdArgs a) {
    bank_body<D, KC, LI, TA>(a, (int)blockIdx.x, (int)gridDim.x);
}

// The four degrees' bank-gradient blocks in ONE launch: block b belongs to segment s (blk_start[s] <= b <
// blk_start[s + 1]) = degree order[s], heaviest degree first; the blocks of the next degree start as soon as a CU is
// free.  133 us alone against 153 us for four launches at batch 4096 (in the replayed graph the step is the same: it is
// bound by the combined work of the two backward chains); at small batches, where a degree has fewer tiles than the
// GPU has CUs and four launches of latency-bound blocks ran one after the other, 4-10 % of a step.
struct BankFusedArgs {
    BwdArgs d[4];
    int blk_start[5];
    int order[4];
    int nseg;
};

template <int KC>
__global__ void __launch_bounds__(512) kc_backward_bank_fused(BankFusedArgs fa) {
    int sgm = 0;
    for (int q = 1; q < fa.nseg; ++q) if ((int)blockIdx.x >= fa.blk_start[q]) sgm = q;
    const int vblock = (int)blockIdx.x - fa.blk_start[sgm], vgrid = fa.blk_start[sgm + 1] - fa.blk_start[sgm];
    switch (fa.order[sgm]) {
        case 0: bank_body<1, KC, bank_li(1), bank_ta(1)>(fa.d[0], vblock, vgrid); break;
        case 1: bank_body<2, KC, bank_li(2), bank_ta(2)>(fa.d[1], vblock, vgrid); break;
        case 2: bank_body<3, KC, bank_li(3), bank_ta(3)>(fa.d[2], vblock, vgrid); break;
        default: bank_body<4, KC, bank_li(4), bank_ta(4)>(fa.d[3], vblock, vgrid); break;
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
    static PerDeviceOnce attr_set;
    if (const int slot = attr_set.pending(); slot >= 0) {
        hipError_t e = hipFuncSetAttribute((const void*)kc_backward_rows_lds<D, KC>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)kc_backward_bank_lds<D, KC, LI, TA>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        attr_set.set(slot);
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
        static const char* env_blocks = getenv("MKGNN_BANK_BLOCKS");          // diagnostics
        int64_t blocks = env_blocks ? atoi(env_blocks) : BWD_BANK_BLOCKS;
        if (blocks < 1 || blocks > BWD_BANK_BLOCKS) blocks = BWD_BANK_BLOCKS;
        // every block ends by writing a full partial slab (115 KB for degree 4): with about one tile per block the
        // slab traffic costs as much as the tile (stamps: 15 us of work, 41 us of kernel), so a block takes at
        // least two tiles
        // (when there are more tiles than blocks; a small batch -- fewer tiles than CUs -- is latency-bound per tile
        // and takes one tile per block: degree 4 at batch 256 has 16 tiles, 36 us on 8 blocks)
        if (ntiles > BWD_BANK_BLOCKS && blocks > (ntiles + 1) / 2) blocks = (ntiles + 1) / 2;
        if (blocks > ntiles) blocks = ntiles;
        if (blocks < 1) blocks = 1;
        a.nchunk = (int)blocks;
        a.theta_in_bank = rows_too ? 0 : 1;          // the LDS rows kernel sums the score-weight partials itself
        if (!rows_too) *ntheta_out = (int)blocks;
        kc_backward_bank_lds<D, KC, LI, TA><<<(int)blocks, 512, lds_bytes, st>>>(a);
        *nchunk_out = (int)blocks;
    }
    return hipGetLastError();
}

static size_t bank_lds_bytes(int d, int FP) {
    const int TA = bank_ta(d), CW = (d == 1) ? 2 : ((d == 4) ? 8 : 4);
    return ((size_t)TA * (d + 1) * (FP + 8) + CW * TA * (size_t)(8 * bank_li(d)) + 4 * TA * (d + 1)) * 4;
}

// blocks the bank kernel of degree d uses for n atoms (the rule of launch_lds_bwd)
int bank_blocks_for(int d, int64_t n) {
    const int64_t ntiles = (n + bank_ta(d) - 1) / bank_ta(d);
    int64_t blocks = BWD_BANK_BLOCKS;
    if (ntiles > BWD_BANK_BLOCKS && blocks > (ntiles + 1) / 2) blocks = (ntiles + 1) / 2;
    if (blocks > ntiles) blocks = ntiles;
    return (int)(blocks < 1 ? 1 : blocks);
}

// One launch for the bank gradients of every degree in `use` (all with the MFMA rows kernel in front: the bank kernel
// sums the score-weight partials).  nchunk_out[i] / ntheta_out[i] = slabs written for degree i + 1.
hipError_t launch_backward_bank_fused(const BwdArgs a4[4], const bool use[4], int nchunk_out[4], int ntheta_out[4], hipStream_t st) {
    BankFusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    int KC = 0;
    size_t lds_bytes = 0;
    int order[4], n_use = 0;
    for (int i = 0; i < 4; ++i) if (use[i]) order[n_use++] = i;
    for (int x = 0; x < n_use; ++x)                  // heaviest first: work ~ atoms * kernels * (d + 1)
        for (int y = x + 1; y < n_use; ++y) {
            const double wx = (double)a4[order[x]].n * a4[order[x]].L * (order[x] + 2), wy = (double)a4[order[y]].n * a4[order[y]].L * (order[y] + 2);
            if (wy > wx) { const int t = order[x]; order[x] = order[y]; order[y] = t; }
        }
    int blk = 0;
    for (int x = 0; x < n_use; ++x) {
        const int i = order[x], d = i + 1;
        fa.d[i] = a4[i];
        const int blocks = bank_blocks_for(d, a4[i].n);
        fa.d[i].nchunk = blocks;
        fa.d[i].theta_in_bank = 1;
        fa.order[x] = i;
        fa.blk_start[x] = blk;
        blk += blocks;
        nchunk_out[i] = blocks;
        ntheta_out[i] = blocks;
        const int FP = mfma_padded_width(a4[i].F);
        KC = FP / 16;
        const size_t b = bank_lds_bytes(d, FP);
        if (b > lds_bytes) lds_bytes = b;
    }
    fa.blk_start[n_use] = blk;
    fa.nseg = n_use;
    if (n_use == 0) return hipSuccess;
    static PerDeviceOnce attr_set[2];
    const int which = KC == 2 ? 0 : 1;
    if (const int slot = attr_set[which].pending(); slot >= 0) {
        hipError_t e = KC == 2 ? hipFuncSetAttribute((const void*)kc_backward_bank_fused<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024)
                               : hipFuncSetAttribute((const void*)kc_backward_bank_fused<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        attr_set[which].set(slot);
    }
    if (KC == 2) kc_backward_bank_fused<2><<<blk, 512, lds_bytes, st>>>(fa);
    else kc_backward_bank_fused<7><<<blk, 512, lds_bytes, st>>>(fa);
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

extern "C" int mkgnn_debug_set_bwd_stamp_buffer(void* device_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mkgnn::g_bwd_stamp_buffer), &device_ptr, sizeof(void*));
}
