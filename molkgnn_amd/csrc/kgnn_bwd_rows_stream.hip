// Streamed MFMA "rows" kernel of the backward pass (gfx950): the gradient with respect to the unit feature rows,
// all four degree buckets in ONE launch (autograd of reference kernels.py:353-425 towards x, before the scatter of
// kernels.py:527, 543 is undone by the gather kernel).
//
//   g_xhat[n, slot a, :] = sum_b sum_l coef[n, l] * [pi_{n,l}(a) = b] * unit_support[l, b, :]
//   g_xhat[n, focal,  :] = sum_l coefc[n, l] * unit_centre[l, :]
//
// Per 16-atom tile and neighbour slot this is P_ab [atoms x kernels] . S_b [kernels x F] with the 0/1-masked
// coefficient tile as the A operand.  kc_backward_rows_mfma (kgnn_bwd_mfma.hip) reads the bank from LDS, one 4-byte
// read per matrix instruction, and runs once per degree.  Here, as in the forward's streamed kernel, the BANK is the
// register-resident operand: a wave owns one column tile (<= 16 kernels) of its degree and keeps those kernels' unit
// rows in B-operand order (kernel k, feature j) in 28 (D + 1) VGPRs for the whole launch; nothing is gathered (the
// product needs no feature rows at all) and the inner loop is matrix instructions only.  A wave's result is the partial
// sum over ITS kernels: the NS waves that hold a degree's NS column tiles exchange the partial tiles through LDS
// (row-major images, laid out like the contribution rows), each adds up a share of the ATOMS in a fixed order and
// stores whole 16-byte chunks of those rows -- bit-reproducible, no float atomics, 4 store instructions per slot
// instead of 14 scattered 4-byte ones.  The coefficient inputs of a tile come from the backward's pre-pass
// (coef_prepare_kernel, kgnn_bwd_stream.hip: dL/dsc times the chirality sign and the permutation ids in tile order, one
// contiguous 2 KB record per (tile, column tile), loaded one tile ahead); without a pre-pass (bank kernel not streamed)
// the kernel gathers them itself through the focal ids, one tile ahead, the focal ids two.
//
// Covered shapes: the streamed forward's (any F <= 112: KC = 1 .. 7 chunks; any number of kernels).  A stream's
// NS(d) = 1 / 2 / 2 / 4 waves hold one column tile each; a degree with more column tiles is done in PASSES (one launch per
// column part, all degrees that still have a part in it): pass 0 writes the contribution rows, every later pass adds its
// kernels' share to them (load, add, store: fixed order, bit-reproducible).  The reference's 10 / 20 / 30 / 50 take one pass.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "kgnn_launch.h"
#include "kgnn_split.h"

// (A/B build: degree 1, whose stream is one wave, stores its chunks straight from the registers instead of going through the
// exchange image -- 64-byte pieces of a row per instruction instead of whole 448-byte rows: measured 62-67 us against 57)
#ifndef MKGNN_ROWS_DIRECT
#define MKGNN_ROWS_DIRECT 0
#endif
#ifndef MKGNN_ROWS_RS_PAD                    // (A/B builds: make VARIANT=rspad0 EXTRA=-DMKGNN_ROWS_RS_PAD=0 is a linear image)
#define MKGNN_ROWS_RS_PAD 4
#endif

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build (make STAMPS=1), as in kgnn_bwd_stream.hip.  Rows kernel phases: 0 tile coefficients + next tile's loads
// issued, 1 matrix products of a slot, 2 partial tile into LDS, 3 barrier, 4 finishing pass (read, add, store).
#ifdef MKGNN_BWD_STAMPS
__device__ unsigned long long* g_rows_stream_stamps = nullptr;
#define MKGNN_RPHASE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); phase[i] += t_ - t_phase; t_phase = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MKGNN_RPHASE(i) do { } while (0)
#endif

namespace rs {

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (B < E) { fn(IC<B>{}); static_for<B + 1, E>(fn); }
}
template <int D, int A> __device__ __forceinline__ int perm_entry(int p) {
    if constexpr (D == 1) return 0;
    else {
        constexpr uint32_t packed = [] {
            uint32_t v = 0;
            for (int q = 0; q < PermC<D>::P; ++q) v |= (uint32_t)PermC<D>::t[q][A] << (2 * q);
            return v;
        }();
        return (int)((packed >> (2 * p)) & 3u);
    }
}
__host__ __device__ constexpr int column_tiles(int d) { return d == 1 ? 1 : (d == 4 ? 4 : 2); }

}  // namespace rs

struct RowsStreamDeg {
    const int64_t* sel;
    const float* pair; const int8_t* chir;
    const float* padded; const float* mix;
    float* contrib; int64_t contrib_base;
    const float* coefq;      // the pre-pass's records [tile][column tile][g 16 x 16 | idx 16 x 16] (null: gather here)
    int64_t n;
    int L, off, kpt, nct;
};

struct RowsStreamArgs {
    const float* gout; int64_t gs;
    int F, CS;
    int FPB;                 // row pitch of the padded bank
    int cp;                  // column part (pass) of this launch: waves hold column tiles cp * NS + role; > 0: add to the rows
    RowsStreamDeg deg[MKGNN_MAX_DEGREE];
    uint8_t grp_degree[4];
    uint16_t grp_count[4];
    uint8_t blk_group[FUSED_MAX_BLOCKS];
    uint16_t blk_rank[FUSED_MAX_BLOCKS];
};

// SP = true (round 5): the products as split fp16 (kgnn_split.h): the masked coefficient tile is the A operand, a lane's four
// values belong to ONE atom (lane & 15) and are scaled by a power of two taken from the atom's largest coefficient; the bank
// rows (register-resident) are split once; the tile is computed TRANSPOSED (bank as A, coefficients as B), so that a lane
// holds sixteen contiguous bytes of its own atom's row: scaled back by the lane's own factor, written to the exchange image
// with one 16-byte LDS write per feature tile (degree 1, a single column tile: straight to memory).  k-position (lane >> 4, i) of the 16-deep product stands for kernel 4 (lane >> 4) + i (see kix below).
constexpr int ROWS_COEF_EXP = 10, ROWS_BANK_EXP = 12;

template <int D, int KC, bool SP>
__device__ __forceinline__ void rows_stream_body(const RowsStreamArgs& a, const RowsStreamDeg& dg, const int rank, const int count,
                                                 float* lds) {
    using namespace rs;
    constexpr int NS = column_tiles(D), NSTREAM = 4 / NS, S1 = D + 1;
    constexpr int FP = 16 * KC;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef MKGNN_BWD_STAMPS
    const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stream = wave / NS, role = wave % NS;
    const int ci = lane & 15, kq = lane >> 4;
    const int L = dg.L, kpt = dg.kpt;
    const int ct = a.cp * NS + role;
    const bool ct_ok = ct < dg.nct;                  // (past the degree's last column tile: an idle wave, zero partial tiles)
    const int ctc = ct_ok ? ct : dg.nct - 1;
    const bool add = a.cp > 0;
    const int FPB = a.FPB;
    // exchange images: [stream][parity][role][atom 16][FP floats] (row-major, like the contribution rows)
    // LDS row stride: FP + 4 floats keeps the 4-byte writes of a quad (rows 4 kq + r) on different banks.  The finishing pass
    // reads 16-byte chunks: a ds_read_b128 is served in four fixed 16-lane groups, each of which tiles the 64 banks when its
    // lanes read consecutive chunks OF ONE ROW -- so a half-wave takes one row (lanes 28..31 of the half idle), whatever the
    // stride.  (Round 2 ran 64 consecutive chunks of the padded image through every instruction: three of the four groups
    // straddled a row's pad chunk and hit a busy bank, 40 % of the kernel's LDS cycles were conflict cycles; a linear image
    // without padding moved the conflicts to the writes instead, 2.2 M conflict cycles of 5.7 M.  Neither changes the
    // kernel's time -- 72.4 / 72.8 us: the exchange is ~4 % of its wave cycles.)
    constexpr int RS = FP + MKGNN_ROWS_RS_PAD;
    float* const xbuf = lds + (size_t)stream * (2 * NS * 16 * RS);

    // (tile bookkeeping in 32 bits -- the streamed kernels take n_atoms * stride < 2^30 only; addresses stay 64-bit)
    const int ntiles = (int)((dg.n + 15) / 16);
    const int nstreams = count * NSTREAM;
    const int sg = rank * NSTREAM + stream;
    const int tile_first = (int)((int64_t)sg * ntiles / nstreams);
    const int tile_end = (int)((int64_t)(sg + 1) * ntiles / nstreams);
    const int iters = (ntiles + nstreams - 1) / nstreams;
    const int tile_hi = (tile_end > tile_first ? tile_end : (tile_first + 1 < ntiles ? tile_first + 1 : ntiles)) - 1;
    auto tile_at = [&](int i) -> int {
        const int t = tile_first + i;
        return t > tile_hi ? tile_hi : t;
    };

    const float w_s = dg.mix[0], w_c = dg.mix[1], w_sum = dg.mix[3];
    const float ws_n = w_s / w_sum / (float)D;
    const float ratio_c = w_c * (float)D / w_s;
    const int8_t* const chp = dg.chir ? dg.chir : (const int8_t*)dg.pair;        // always loadable; ignored without signs

    // coefficient inputs of a tile for this lane: atom ci, kernels kix(q) of the wave's column tile (clamped, masked later).
    // fp32 instructions: k-step q holds kernel 4 q + kq.  Split products: the labelling of the 16-deep product's k-positions is
    // free, and (kq, q) <-> kernel 4 kq + q makes the lane's four coefficients (and ids) ONE 16-byte load of the record row.
    auto kix = [&](int q) -> int { return SP ? 4 * kq + q : 4 * q + kq; };
    float rg[4];
    int ridx[4], rch[4];
    auto focal_of = [&](int64_t tile) -> int64_t {
        const int64_t n = tile * 16 + ci;
        return dg.sel[n < dg.n ? n : dg.n - 1];
    };
    // (the split body is launched with the pre-pass's records only: launch_rows_pass -- no gather path, fewer registers)
    const bool records = SP ? true : (dg.coefq != nullptr);
    auto issue = [&](int64_t tile, int64_t focal) {
        if (records) {
            // the pre-pass (coef_prepare_kernel) has put dL/dsc (signed, zero for padding) and the permutation ids into
            // tile order: two contiguous 1 KB images per (tile, column tile), this lane's entries at atom ci, kernel 4 q + kq
            if constexpr (SP) {
                const float* rec = dg.coefq + ((size_t)tile * dg.nct + ctc) * 512 + ci * 16 + 4 * kq;
                const f32x4 g4 = *(const f32x4*)rec, i4 = *(const f32x4*)(rec + 256);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rg[q] = g4[q];
                    ridx[q] = __float_as_int(i4[q]) & 0xff;  // (above the id: the record's largest exponent, for the bank kernel)
                    rch[q] = 1;
                }
                return;
            }
            const float* rec = dg.coefq + ((size_t)tile * dg.nct + ctc) * 512 + ci * 16 + kq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rg[q] = rec[4 * q];
                ridx[q] = __float_as_int(rec[256 + 4 * q]) & 0xff;
                rch[q] = 1;
            }
            return;
        }
        const int64_t n = tile * 16 + ci;
        const int64_t nc = n < dg.n ? n : dg.n - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int l = ctc * kpt + kix(q);
            const int lc = l < L ? l : L - 1;
            rg[q] = a.gout[focal * a.gs + dg.off + lc];
            ridx[q] = pair_index(dg.pair, (size_t)nc * L + lc);
            rch[q] = chp[(size_t)nc * L + lc];
        }
    };
    issue(tile_at(0), records ? 0 : focal_of(tile_at(0)));
    int64_t focal_next = records ? 0 : focal_of(tile_at(1));
    int par = 0;

    // (the first tile's coefficient loads are in flight while the bank is loaded: one dependent round trip less in front of
    // the first tile)
    // ---- one-time: this wave's kernels' unit rows in B-operand order: lane (k = kq, j = ci) -> kernel 4 q + kq, feature 16 t + ci
    using BankT = std::conditional_t<SP, SplitReg, f32x4>;
    BankT bk[D + 1][KC];                                  // [.][t]: kernels 4 q + kq, q = 0..3 (fp32: component q)
    {
        bool ok[4];
        int lc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = kix(q), l = ct * kpt + i;
            ok[q] = ct_ok && i < kpt && l < L;
            lc[q] = ok[q] ? l : 0;
        }
#pragma unroll
        for (int b = 0; b <= D; ++b)
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float e = dg.padded[((size_t)b * L + lc[q]) * FPB + 16 * t + ci];
                    v[q] = ok[q] ? e : 0.f;
                }
                if constexpr (SP) bk[b][t] = split_scaled(v, (float)(1 << ROWS_BANK_EXP));
                else bk[b][t] = v;
            }
    }

#ifdef MKGNN_BWD_STAMPS
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_phase = t_start;
    phase[5] = t_start - t_entry;                        // the prologue: the bank into registers, tile 0's coefficient loads issued
#endif
    for (int it = 0; it < iters; ++it) {
        const int64_t tile = tile_at(it);
        const bool real = tile_first + it < tile_end;
        const int64_t n_mine = tile * 16 + ci;
        float cf[4];
        int ix[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = kix(q);
            const bool ok = real && ct_ok && n_mine < dg.n && i < kpt && ct * kpt + i < L;
            const float g = (dg.chir && !records) ? rg[q] * (float)rch[q] : rg[q];
            cf[q] = ok ? g * ws_n : 0.f;
            ix[q] = ridx[q];
        }
        // next tile's inputs: in flight during this tile's matrix work
        issue(tile_at(it + 1), focal_next);
        if (!records) focal_next = focal_of(tile_at(it + 2));
        // (SP) the atom's scale: its largest coefficient (the centre's ratio included) to [2^10, 2^11); the result the lane
        // holds is the same atom's (transposed tile, below), so it undoes its own scale
        [[maybe_unused]] float sca = 1.f, unsc = 1.f;
        if constexpr (SP) {
            float am = fmaxf(fmaxf(fabsf(cf[0]), fabsf(cf[1])), fmaxf(fabsf(cf[2]), fabsf(cf[3])));
            am = fmaxf(am, __shfl_xor(am, 16));
            am = fmaxf(am, __shfl_xor(am, 32));
            am *= fmaxf(1.f, fabsf(ratio_c));
            sca = split_scale_for<ROWS_COEF_EXP>(am);
            unsc = split_unscale_of<ROWS_BANK_EXP>(sca);
        }
        MKGNN_RPHASE(0);

        static_for<0, S1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;       // contribution-row slot: 0 = focal (centre rows), 1 + a = neighbour a
            f32x4 acc[KC];
#pragma unroll
            for (int t = 0; t < KC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) {
                // The TRANSPOSED tile: the bank registers as the A operand (feature 16 t + ci x kernels), the coefficients as B
                // (kernels x atom ci) -- the same registers either way round -- so that a lane ends up with features
                // 16 t + 4 kq .. + 3 of atom ci: sixteen contiguous bytes of that atom's contribution row, and the atom's scale is
                // the lane's own.  Three instructions per feature tile, KC independent chains.
                auto multiply = [&](const SplitReg& av, const SplitReg* row) {
#pragma unroll
                    for (int t = 0; t < KC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(row[t].hi, av.lo, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < KC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(row[t].lo, av.hi, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < KC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(row[t].hi, av.hi, acc[t], 0, 0, 0);
                };
                if constexpr (s == 0) {
                    multiply(split_scaled(f32x4{cf[0] * ratio_c, cf[1] * ratio_c, cf[2] * ratio_c, cf[3] * ratio_c}, sca), bk[D]);
                } else {
                    int pb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) pb[q] = perm_entry<D, (s > 0 ? s - 1 : 0)>(ix[q]);
#pragma unroll
                    for (int b = 0; b < D; ++b)
                        multiply(split_scaled(f32x4{pb[0] == b ? cf[0] : 0.f, pb[1] == b ? cf[1] : 0.f, pb[2] == b ? cf[2] : 0.f,
                                                    pb[3] == b ? cf[3] : 0.f}, sca), bk[b]);
                }
#pragma unroll
                for (int t = 0; t < KC; ++t) acc[t] *= unsc;
            } else if constexpr (s == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float av = cf[q] * ratio_c;
#pragma unroll
                    for (int t = 0; t < KC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bk[D][t][q], acc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int pb = perm_entry<D, (s > 0 ? s - 1 : 0)>(ix[q]);
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        const float av = (pb == b) ? cf[q] : 0.f;
#pragma unroll
                        for (int t = 0; t < KC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bk[b][t][q], acc[t], 0, 0, 0);
                    }
                }
            }
            // ---- sum over the column tiles (waves of the stream) and write the contribution rows.  A scattered 4-byte
            // store instruction costs a wave 100 - 250 cycles here (measured in the forward), and the product would issue
            // 14 of them per slot: instead every wave writes its partial tile into LDS as ROWS ([atom][FP] floats, the
            // layout of the contribution rows themselves), and after the barrier each wave adds up the NS images of its
            // share of the atoms 16 bytes at a time and stores whole 16-byte chunks of contiguous 448-byte rows.
            MKGNN_RPHASE(1);
            float* const mine = xbuf + (size_t)((par * NS + role) * 16) * RS;
            if constexpr (SP && NS == 1 && MKGNN_ROWS_DIRECT) {
                // one column tile per degree: the lane's chunks ARE the contribution row's -- straight to memory, no exchange
                const int64_t nn = tile * 16 + ci;
                if (real && nn < dg.n) {
                    float* const row = dg.contrib + (size_t)(dg.contrib_base + nn * S1 + s) * a.CS + 4 * kq;
                    const int F4 = (a.F + 3) / 4;
#pragma unroll
                    for (int t = 0; t < KC; ++t)
                        if (4 * t + kq < F4) {
                            f32x4 v = acc[t];
                            if (add) v += *(const f32x4*)(row + 16 * t);
                            *(f32x4*)(row + 16 * t) = v;
                        }
                }
            } else if constexpr (SP) {
#pragma unroll
                for (int t = 0; t < KC; ++t) *(f32x4*)(mine + ci * RS + 16 * t + 4 * kq) = acc[t];
            } else {
#pragma unroll
                for (int t = 0; t < KC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[(kq * 4 + r) * RS + 16 * t + ci] = acc[t][r];
            }
            MKGNN_RPHASE(2);
            if constexpr (NS > 1) __syncthreads();
            else __builtin_amdgcn_wave_barrier();        // (own image: LDS accesses of one wave are ordered)
            MKGNN_RPHASE(3);
            if constexpr (!(SP && NS == 1 && MKGNN_ROWS_DIRECT)) {
                constexpr int APW = 16 / NS;             // atoms this wave finishes
                constexpr int CPR = FP / 4;              // 16-byte chunks per row
                static_assert(CPR <= 64, "at most one wave per row");
                const float* const img = xbuf + (size_t)(par * NS * 16) * RS;
                const int F4 = (a.F + 3) / 4;            // chunks that hold row data (CS = F rounded up to 4: the row's own padding)
                // (rows of more than 32 chunks, KC > 8: a whole wave per row; narrow rows -- the 1-hop layer's 28 floats are 7
                // chunks -- share an instruction eight or four at a time: a finishing pass of 1 instead of 4 read-add-store
                // rounds per slot)
                constexpr int LPRR = CPR <= 8 ? 8 : (CPR <= 16 ? 16 : (CPR <= 32 ? 32 : 64));   // lanes per row
                constexpr int RPI = 64 / LPRR;           // rows per wave-instruction
                const int ch = lane % LPRR, hw = lane / LPRR;
#pragma unroll
                for (int a0 = 0; a0 < APW; a0 += RPI) {
                    const int al = a0 + hw;
                    const int atom = role * APW + (al < APW ? al : 0);
                    if (al < APW && ch < F4 && ch < CPR) {
                        f32x4 v = *(const f32x4*)(img + (size_t)atom * RS + 4 * ch);
#pragma unroll
                        for (int w = 1; w < NS; ++w) v += *(const f32x4*)(img + (size_t)(w * 16 + atom) * RS + 4 * ch);
                        const int64_t nn = tile * 16 + atom;
                        if (real && nn < dg.n) {
                            f32x4* const dst = (f32x4*)(dg.contrib + (size_t)(dg.contrib_base + nn * S1 + s) * a.CS + 4 * ch);
                            if (add) v += *dst;              // (a later pass: this column part's share on top of the earlier ones')
#ifndef MKGNN_NO_NT_STORE                                 // streaming stores: the rows are written once here and read once by the gather
                            __builtin_nontemporal_store(v, dst);     // (52.7 -> 50.5 us alone, the step -0.5 %; -DMKGNN_NO_NT_STORE: A/B)
#else
                            *dst = v;
#endif
                        }
                    }
                }
            }
            par ^= 1;
            MKGNN_RPHASE(4);
        });
    }
#ifdef MKGNN_BWD_STAMPS
    if (g_rows_stream_stamps && lane == 0) {
        unsigned long long* o = g_rows_stream_stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
        o[0] = t_start; o[1] = __builtin_readcyclecounter(); o[2] = (unsigned long long)(D * 16 + a.cp); o[3] = (unsigned long long)iters;
        for (int i = 0; i < 8; ++i) o[4 + i] = phase[i];
    }
#endif
}

template <int KC, bool SP = false>
__global__ void __launch_bounds__(256, (KC >= 8 ? 1 : 2)) kc_backward_rows_stream(RowsStreamArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int grp = a.blk_group[blockIdx.x];
    const int rank = a.blk_rank[blockIdx.x];
    const int di = a.grp_degree[grp];
    const int count = a.grp_count[grp];
    switch (di) {
        case 0: rows_stream_body<1, KC, SP>(a, a.deg[0], rank, count, lds); break;
        case 1: rows_stream_body<2, KC, SP>(a, a.deg[1], rank, count, lds); break;
        case 2: rows_stream_body<3, KC, SP>(a, a.deg[2], rank, count, lds); break;
        default: rows_stream_body<4, KC, SP>(a, a.deg[3], rank, count, lds); break;
    }
}

// ---------------------------------------------------------------- host ----
extern "C" int mkgnn_debug_set_rows_stream_stamps(void* device_ptr) {
#ifdef MKGNN_BWD_STAMPS
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stream_stamps), &device_ptr, sizeof(void*));
#else
    (void)device_ptr;
    return -1;
#endif
}

bool rows_stream_supported(int d, int F, int E, int L) {
    if (d < 1 || d > 4 || L < 1) return false;
    if (F < 1 || F > STREAM_MAX_F || !bank_pitch(F)) return false;
    return (L + 15) / 16 <= 8 * rs::column_tiles(d);      // (at most eight passes)
}

template <int KC, bool SP = false> static hipError_t launch_rows_kc(const RowsStreamArgs& a, int nb, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 64 * 1024) {
        static PerDeviceOnce attr_set;
        if (const int slot = attr_set.pending(); slot >= 0) {
            hipError_t e = hipFuncSetAttribute((const void*)kc_backward_rows_stream<KC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (e != hipSuccess) return e;
            attr_set.set(slot);
        }
    }
    kc_backward_rows_stream<KC, SP><<<nb, 256, lds_bytes, st>>>(a);
    return hipGetLastError();
}

// MKGNN_BWD_SPLIT: 1 = the backward's products as split fp16 (kgnn_split.h), 0 = v_mfma_f32_16x16x4_f32, unset = default;
// mkgnn_debug_set_backward_products overrides (tests)
#ifndef MKGNN_BWD_SPLIT_DEFAULT
#define MKGNN_BWD_SPLIT_DEFAULT 1
#endif
static std::atomic<int> g_bwd_split_override{-1};
int bwd_split_mode() {
    static const int m = [] { const char* e = getenv("MKGNN_BWD_SPLIT"); return e ? atoi(e) : MKGNN_BWD_SPLIT_DEFAULT; }();
    const int o = g_bwd_split_override.load(std::memory_order_relaxed);
    return o >= 0 ? o : m;
}
extern "C" int mkgnn_debug_set_backward_products(int32_t mode) { g_bwd_split_override.store(mode < 0 ? -1 : (mode ? 1 : 0)); return 0; }

// one pass (column part cp) over the degrees in `use`
static hipError_t launch_rows_pass(const BwdArgs a4[4], const bool use[4], float* const coefq[4], int cp, hipStream_t st) {
    RowsStreamArgs a;
    memset(&a, 0, sizeof(a));
    int KC = 0, ng = 0;
    double cost[4];
    int64_t tiles_of[4], cap[4];
    int nstream_of[4];
    for (int i = 0; i < 4; ++i) {
        if (!use[i]) continue;
        const BwdArgs& s = a4[i];
        const int d = i + 1;
        a.gout = s.gout; a.gs = s.gs; a.F = s.F; a.CS = s.CS;
        a.FPB = bank_pitch(s.F); a.cp = cp;
        KC = (s.F + 15) / 16;
        RowsStreamDeg& g = a.deg[i];
        g.sel = s.sel; g.pair = s.pair; g.chir = s.chir; g.padded = s.padded; g.mix = s.mix;
        g.contrib = s.contrib; g.contrib_base = s.contrib_base;
        g.coefq = coefq ? coefq[i] : nullptr;
        g.n = s.n; g.L = s.L; g.off = s.off;
        g.nct = (s.L + 15) / 16;
        g.kpt = (s.L + g.nct - 1) / g.nct;
        const int nstream = 4 / rs::column_tiles(d);
        const int64_t ntiles = (s.n + 15) / 16;
        // a wave's time per tile (units of 32 cycles): matrix instructions + per-slot exchange and stores
        cost[ng] = (d * d + 1) * 4.0 * KC + (d + 1) * 30.0 + 40.0;
        if (KC <= 7 && bwd_split_mode() != 0) {
            // the split-fp16 products: measured per tile (tools/bwd_stream_stamps.py, batch 4096, two waves per SIMD) at F = 110
            // and F = 28, other chunk counts interpolated
            static const double sp7[4] = {216.0, 320.0, 462.0, 769.0}, sp2[4] = {106.0, 173.0, 220.0, 310.0};
            cost[ng] = sp2[i] + (sp7[i] - sp2[i]) * (KC - 2) / 5.0;
        }
        tiles_of[ng] = ntiles;
        cap[ng] = (ntiles + nstream - 1) / nstream;
        nstream_of[ng] = nstream;
        a.grp_degree[ng] = (uint8_t)i;
        ++ng;
    }
    if (ng == 0) return hipSuccess;
    auto finish = [&](int g, int blocks) {
        const int64_t streams = (int64_t)blocks * nstream_of[g];
        return 40.0 + (double)((tiles_of[g] + streams - 1) / streams) * cost[g];
    };
    int count[4], nb = 0;
    for (int g = 0; g < ng; ++g) { count[g] = 1; ++nb; }
    // grid cap: see launch_backward_bank_stream (the two kernels' caps were measured together)
    static const char* env_blocks = getenv("MKGNN_ROWS_STREAM_BLOCKS");
    const int max_blocks = grid_cap(g_grid_caps.rows, env_blocks && atoi(env_blocks) > 8 && atoi(env_blocks) <= FUSED_MAX_BLOCKS ? atoi(env_blocks) : 448);
    while (nb < max_blocks) {
        int worst = -1;
        double t_worst = -1.0;
        for (int g = 0; g < ng; ++g) {
            if (count[g] >= cap[g]) continue;
            const double t = finish(g, count[g]);
            if (t > t_worst) { t_worst = t; worst = g; }
        }
        if (worst < 0) break;
        ++count[worst]; ++nb;
    }
    int given[4] = {0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        int pick = -1;
        double best = -1e30;
        for (int g = 0; g < ng; ++g) {
            if (given[g] >= count[g]) continue;
            const double lag = (double)count[g] * (b + 1) / nb - given[g];
            if (lag > best) { best = lag; pick = g; }
        }
        a.blk_group[b] = (uint8_t)pick;
        ++given[pick];
    }
    {
        int per_xcd[4][8] = {};
        for (int b = 0; b < nb; ++b) ++per_xcd[a.blk_group[b]][b & 7];
        int next[4][8];
        for (int g = 0; g < ng; ++g) {
            int run = 0;
            for (int x = 0; x < 8; ++x) { next[g][x] = run; run += per_xcd[g][x]; }
        }
        for (int b = 0; b < nb; ++b) a.blk_rank[b] = (uint16_t)next[a.blk_group[b]][b & 7]++;
    }
    for (int g = 0; g < ng; ++g) a.grp_count[g] = (uint16_t)count[g];
    if (cp == 0) note_plan(1, nb, ng, tiles_of, count, nstream_of);
    g_last_plan[1].launches.fetch_add(1);
    // NSTREAM * NS = 4 wave images of 16 rows, two parities
    const size_t lds_bytes = (size_t)4 * 2 * 16 * (16 * KC + MKGNN_ROWS_RS_PAD) * 4;
    if (KC <= 7 && bwd_split_mode() != 0 && coefq) {
        switch (KC) {
            case 1: return launch_rows_kc<1, true>(a, nb, lds_bytes, st);
            case 2: return launch_rows_kc<2, true>(a, nb, lds_bytes, st);
            case 3: return launch_rows_kc<3, true>(a, nb, lds_bytes, st);
            case 4: return launch_rows_kc<4, true>(a, nb, lds_bytes, st);
            case 5: return launch_rows_kc<5, true>(a, nb, lds_bytes, st);
            case 6: return launch_rows_kc<6, true>(a, nb, lds_bytes, st);
            default: return launch_rows_kc<7, true>(a, nb, lds_bytes, st);
        }
    }
    switch (KC) {
        case 1: return launch_rows_kc<1>(a, nb, lds_bytes, st);
        case 2: return launch_rows_kc<2>(a, nb, lds_bytes, st);
        case 3: return launch_rows_kc<3>(a, nb, lds_bytes, st);
        case 4: return launch_rows_kc<4>(a, nb, lds_bytes, st);
        case 5: return launch_rows_kc<5>(a, nb, lds_bytes, st);
        case 6: return launch_rows_kc<6>(a, nb, lds_bytes, st);
        case 7: return launch_rows_kc<7>(a, nb, lds_bytes, st);
        case 8: return launch_rows_kc<8>(a, nb, lds_bytes, st);
        case 9: return launch_rows_kc<9>(a, nb, lds_bytes, st);
        case 10: return launch_rows_kc<10>(a, nb, lds_bytes, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_backward_rows_stream(const BwdArgs a4[4], const bool use[4], float* const coefq[4], hipStream_t st) {
    int passes = 0;
    for (int i = 0; i < 4; ++i)
        if (use[i]) {
            const int p = ((a4[i].L + 15) / 16 + rs::column_tiles(i + 1) - 1) / rs::column_tiles(i + 1);
            if (p > passes) passes = p;
        }
    for (int cp = 0; cp < passes; ++cp) {
        bool use_p[4];
        for (int i = 0; i < 4; ++i)
            use_p[i] = use[i] && cp * rs::column_tiles(i + 1) < (a4[i].L + 15) / 16;
        hipError_t e = launch_rows_pass(a4, use_p, coefq, cp, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mkgnn
