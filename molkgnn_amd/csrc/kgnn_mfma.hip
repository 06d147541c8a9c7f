// Fused MFMA forward of the kernel convolution: ONE launch computes a whole KernelSetConv
// forward (all four degree buckets) on gfx950 with the exact-fp32 v_mfma_f32_16x16x4_f32.
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
// the bond-cosine matrix IN REGISTERS: the permutation maximum needs no cross-lane traffic.
//
// Work split.  A block belongs to one (degree, column part) group and keeps only ITS share of the
// unit-normalised kernel bank in LDS (<= 57 KB, so two blocks and 2 waves per SIMD fit on a CU
// whatever the degree).  The host sizes the groups in proportion to their MFMA work and
// interleaves them over the block ids, so every CU carries a mix of degrees and all buckets
// start -- and finish -- together.  Inside a block there is no synchronisation after the bank
// copy: every wave walks its own 16-atom tiles.
//
// Slot-pipelined gather.  A wave needs the rows of slot s (neighbour s of its 16 atoms; the focal
// row is slot d) only while it multiplies slot s.  It therefore holds just two slots of A
// fragments: while slot s is multiplied (d * 28 * nloc MFMAs), the 16-byte chunks 4t + kq of
// slot s+1 -- or of the next tile's slot 0 -- are already in flight straight from global memory
// into the other register set (lane (row i, k-quarter kq) loads exactly its MFMA A operand).
// The gather is thus hidden behind the matrix pipe inside each wave, at 56 VGPRs.
// Rows stay RAW; 1/|x| multiplies the dot products.  The bank's 16-byte LDS chunks are
// XOR-swizzled so the ds_read_b128 fragment reads are conflict-free (brute-forced against the
// lane groups of MI355X_MICROARCH.md, LDS section).
#include "kgnn_launch.h"

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic time stamps (tools/stamp_probe.py): per wave {start, bank ready, then per tile
// start / multiplied / stored}.  Null in normal runs.
__device__ unsigned long long* g_stamp_buffer = nullptr;
// Compiled in only with -DMKGNN_FWD_STAMPS (make STAMPS=1): the conditional stores split the tile loop
// into more basic blocks and perturb the wait-count placement of the production kernel.
#ifdef MKGNN_FWD_STAMPS
#define MKGNN_STAMP(slot)                                                                      \
    do {                                                                                       \
        if (stamps && lane == 0 && (slot) < 29) stamps[(slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define MKGNN_STAMP(slot) do { (void)stamps; (void)(slot); } while (0)
#endif

template <int KC> __device__ __forceinline__ int swz(int pos) {
    if constexpr (KC == 7) {                // 112-float rows: rows 4 apart share banks
        return (0x1230 >> (((pos >> 2) & 3) * 4)) & 3;     // {0, 3, 2, 1}[(pos >> 2) & 3]
    } else {                                // 32-float rows: rows 2 apart share banks
        return (pos >> 1) & 7;
    }
}

// The bf16 variant keeps the bank in LDS as bf16 (8-byte chunks, read with ds_read_b64).  Rows are 56 dwords apart
// for 112 columns (rows 4 apart share banks): XOR-ing bit 1 of the chunk index with bit 2 of the row spreads a
// half-wave's 64 dwords evenly over the 32 banks; 32-column rows are 16 dwords apart and take the fp32 swizzle.
template <int KC> __device__ __forceinline__ int swz_h(int pos) {
    if constexpr (KC == 7) return ((pos >> 2) & 1) * 2;
    else return (pos >> 1) & 7;
}

// column tiles one WAVE accumulates at a time (bounds the accumulator registers)
template <int D> struct BodyTraits {
    static constexpr int NL = (D >= 3) ? 1 : 2;
    // Row buffers.  2: the whole following slot is fetched into the other buffer when a slot starts (prefetch
    // distance 1 - 2 slots of matrix work).  1: a single buffer rotated in place, chunk t of the following
    // slot is loaded right after chunk t's MFMAs are issued (distance exactly 1 slot, KC * 4 fewer VGPRs).
    // Degree 4 takes the single buffer: with two it spilled, and every scratch reload carries a vmcnt(0) that
    // also waits for the gather.  For degrees 1-3 both fit; the launch time is the same within noise
    // (77.7 - 78.9 us for all-1 / all-2 / mixed at batch 4096), two buffers keep the longer prefetch distance.
    static constexpr int NB = (D == 4) ? 1 : 2;
};

// LDS floats a (degree, column part) block needs.
__host__ __device__ static inline int fused_lds_floats(int D, int KC, int L, int nloc, int kpt) {
    const int srow = nloc * kpt;
    return (D + 1) * srow * 16 * KC + D * srow * 8 + ((D == 4 ? L * 12 : 0) + 15) / 16 * 4 + 4 * 16 * 8;
}

// Four fp32 values -> four bf16 (round to nearest even), the operand of v_mfma_f32_16x16x16_bf16: a lane's four
// consecutive columns of a chunk are exactly that instruction's k = 4 (lane >> 4) + i.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {
    // (plain conversions: the compiler emits v_cvt_pk_bf16_f32 and knows the VALU -> MFMA operand hazards; the same
    // instruction in inline asm is opaque to its hazard recogniser and produced NaNs)
    const bf16x4 r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(s16x4, r);
}

// GEN = false: only the last 16-float chunk of a row can be partial (FP - 16 < F <= FP: the reference's 28 and 110);
// GEN = true: any F <= FP, every chunk is masked against the row width (a few % more VALU work).
// BF = true: the "bf16 similarity path" (BASELINE configs[4]): node-feature dot products with
// bf16 operands and fp32 accumulation (one 16x16x16 MFMA per chunk instead of four 16x16x4); norms, bond cosines,
// mixing and everything the backward uses stay fp32.  Scores differ from the fp32 path by ~1e-3 and the chosen
// permutation may differ where two orders are that close.
template <int D, int KC, bool GEN, bool BF>
__device__ __forceinline__ void forward_body(const FusedFwdArgs& a, const FusedDeg& dg, const int cp, const int rank,
                                             const int count, float* lds) {
    constexpr int FP = 16 * KC;
    constexpr int CH = 4 * KC;                       // 16-byte chunks per row
    constexpr int NW = 4;                            // waves per block
    constexpr int NL = BodyTraits<D>::NL;            // column tiles per block (static bound)
    const int L = dg.L, kpt = dg.kpt, cs = dg.cs, nloc = dg.nloc, nct = dg.nct;
    const int SROW = nloc * kpt;                     // bank rows per slot in this block
    float* bank = lds;                               // [(D+1)][SROW][FP], slot D = centre rows
    float* esup = bank + (size_t)(D + 1) * SROW * FP;   // [D][SROW][8]
    int8_t* chirtab = (int8_t*)(esup + (size_t)D * SROW * 8);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // per-wave scratch [8][16]: 1/|x| of slots 0..D, focal id, chirality sign, "rows equal" flag of the
    // wave's 16 atoms, written by the row-owning lanes and read back per accumulator row as one b128
    float* wscr = (float*)(chirtab + ((D == 4 ? L * 12 : 0) + 15) / 16 * 16) + wave * 16 * 8;
    const bool do_chir = (D == 4) && a.last;
    unsigned long long* stamps = g_stamp_buffer ? g_stamp_buffer + ((size_t)blockIdx.x * NW + wave) * 32 : nullptr;
    int stamp_slot = 2;
    MKGNN_STAMP(0);
#ifdef MKGNN_FWD_STAMPS
    if (stamps && lane == 0) stamps[31] = (unsigned long long)(D * 16 + cp);
#endif

    const int ci = lane & 15, kq = lane >> 4;
    const int64_t ntiles = (dg.n + 15) / 16;
    // `ics` waves of the block share an atom tile (each takes its own column tiles; the second
    // wave's row loads hit the lines the first one brought in)
    const int ics = dg.ics;
    const int wpart = wave % ics;
    const int tiles_per_block = NW / ics;
    // A wave owns a CONTIGUOUS run of its group's tiles, and the host numbers the blocks of a group so that the
    // blocks sharing an XCD (block id mod 8) own adjacent runs: the buckets are sorted by atom id, so XCD x works on
    // roughly the x-th eighth of the atoms in every degree group at once and the rows one group gathers as
    // neighbours are the rows another group on the same L2 gathers as focal atoms.
    const int64_t nwaves_g = (int64_t)count * tiles_per_block;
    const int64_t wave_g = (int64_t)rank * tiles_per_block + wave / ics;
    int64_t tile = wave_g * ntiles / nwaves_g;
    const int64_t tile_end = (wave_g + 1) * ntiles / nwaves_g;
    const uint32_t xs = (uint32_t)a.xs;

    uint32_t ids[D + 1];                             // this tile: slots 0..D-1 neighbours, slot D focal
    auto load_ids = [&](int64_t t, uint32_t (&dst)[D + 1]) {
        int64_t n = t * 16 + ci;
        if (n >= dg.n) n = dg.n - 1;
#pragma unroll
        for (int s = 0; s < D; ++s) dst[s] = (uint32_t)dg.nei[n * D + s];
        dst[D] = (uint32_t)dg.sel[n];
    };
    bool have = tile < tile_end;
    if (have) load_ids(tile, ids);

    // ---- one-time: this block's kernel rows -> LDS (chunks swizzled by the row's place in its tile)
    for (int base = 0; base < (D + 1) * SROW * CH; base += 256 * 8) {
        f32x4 tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = base + tid + 256 * k;
            tmp[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < (D + 1) * SROW * CH) {
                const int row = q / CH, c = q - row * CH;
                const int s = row / SROW, jr = row - s * SROW;
                const int j = jr / kpt, r = jr - j * kpt;
                const int l = (cp + j * cs) * kpt + r;
                if (l < L && cp + j * cs < nct) tmp[k] = *(const f32x4*)(dg.padded + ((size_t)s * L + l) * FP + 4 * c);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = base + tid + 256 * k;
            if (q < (D + 1) * SROW * CH) {
                const int row = q / CH, c = q - row * CH;
                const int r = (row % SROW) % kpt;
                if constexpr (BF) *(s16x4*)((short*)bank + (size_t)row * FP + 4 * (c ^ swz_h<KC>(r))) = to_bf16x4(tmp[k]);
                else *(f32x4*)(bank + (size_t)row * FP + 4 * (c ^ swz<KC>(r))) = tmp[k];
            }
        }
    }
    for (int q = tid; q < D * SROW * 2; q += 256) {
        const int row = q >> 1, h = q & 1;
        const int s = row / SROW, jr = row - s * SROW;
        const int j = jr / kpt, r = jr - j * kpt;
        const int l = (cp + j * cs) * kpt + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (l < L && cp + j * cs < nct) v = *(const f32x4*)(dg.edge_padded + ((size_t)s * L + l) * 8 + 4 * h);
        *(f32x4*)(esup + (size_t)row * 8 + 4 * h) = v;
    }
    if constexpr (D == 4) {
        if (do_chir)
            for (int q = tid; q < L * 12; q += 256) chirtab[q] = dg.chir[q];
    }
    __syncthreads();
    MKGNN_STAMP(1);
    if (!have) return;

    const float ws = dg.mix[0], wc = dg.mix[1], we = dg.mix[2], wsum = dg.mix[3];

    // ---- pipeline registers
    // ONE row buffer, rotated in place: as soon as the MFMAs of chunk t of a slot are issued, chunk t of the
    // following slot (the next neighbour, the focal row, or slot 0 of the wave's next tile) is loaded into
    // the same registers.  Every chunk is therefore in flight for exactly one slot's worth of matrix work --
    // the same prefetch distance as two alternating buffers -- with KC * 4 fewer VGPRs; with two buffers the
    // degree 3 / 4 paths spilled, and every scratch reload carries a vmcnt(0) that also waits for the gather.
    // All loads are unconditional (clamped tile index past the end): a load inside a conditional block makes
    // the compiler wait for it at the end of the block.
    constexpr int NB = BodyTraits<D>::NB;
    f32x4 rb[NB][KC];
    float inv_cur = 0.f;
    float2 eraw[D];                                  // bond components 2kq, 2kq+1 of (atom ci, slot s)
    int eq_raw = 0, sign_raw = 0;
    const int8_t* eqp = do_chir ? (const int8_t*)dg.eqflag : (const int8_t*)dg.sel;       // always loadable
    const int8_t* sgp = do_chir ? (const int8_t*)dg.signflag : (const int8_t*)dg.sel;
    auto issue_small = [&](int64_t t) {
        int64_t nrow = t * 16 + ci;
        if (nrow >= dg.n) nrow = dg.n - 1;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const float* e = dg.e_nei + (nrow * D + s) * a.E;
            eraw[s].x = e[2 * kq < a.E ? 2 * kq : a.E - 1];         // masked where they are used
            eraw[s].y = e[2 * kq + 1 < a.E ? 2 * kq + 1 : a.E - 1];
        }
        if constexpr (D == 4) {
            eq_raw = eqp[nrow];
            sign_raw = sgp[nrow];
        }
    };
    auto row_of = [&](uint32_t id) -> const float* { return a.x + (id * xs + 4u * kq); };   // 32-bit offset (host checks N * stride < 2^32)
    // chunk t of a lane covers columns 16 t + 4 kq .. + 3; a chunk that starts beyond the row's width is fetched
    // from the row's first columns instead (the load stays unconditional and in bounds) and masked to zero
    auto load_chunk = [&](const float* row, int t) -> f32x4 {
        const int col = 16 * t + 4 * kq;
        const bool maybe_out = GEN || t == KC - 1;
        const int off = (!maybe_out || col < a.F) ? 16 * t : -4 * kq;
        return *(const f32x4*)(row + off);
    };
    auto mask_chunk = [&](f32x4 v, int t) -> f32x4 {
        if (!(GEN || t == KC - 1)) return v;
        const int col = 16 * t + 4 * kq;
        if (col >= a.F) v.x = 0.f;
        if (col + 1 >= a.F) v.y = 0.f;
        if (col + 2 >= a.F) v.z = 0.f;
        if (col + 3 >= a.F) v.w = 0.f;
        return v;
    };
    issue_small(tile);
    {
        const float* row = row_of(ids[0]);
#pragma unroll
        for (int t = 0; t < KC; ++t) rb[0][t] = load_chunk(row, t);
        inv_cur = a.inv[ids[0]];
    }

    for (;;) {
        MKGNN_STAMP(stamp_slot);
        const int64_t nxt_tile = tile + 1;
        const bool have_next = nxt_tile < tile_end;
        uint32_t ids_n[D + 1];
        load_ids(have_next ? nxt_tile : tile, ids_n);
        // ---- per-tile small values (their loads were issued ahead of the rows)
        float2 eu[D];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const float ex = 2 * kq < a.E ? eraw[s].x : 0.f, ey = 2 * kq + 1 < a.E ? eraw[s].y : 0.f;
            float s2 = fmaf(ey, ey, ex * ex);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
            eu[s] = float2{ex * ie, ey * ie};
        }
        float sign_row = 0.f;
        int eq_row = 0;
        if constexpr (D == 4) {
            sign_row = do_chir ? (float)sign_raw : 0.f;
            eq_row = do_chir ? eq_raw : 0;
        }
        const uint32_t focal_row = ids[D];
        float inv_keep[D + 1];

        // ---- accumulate slot by slot; the following slot's rows stream in behind the chunks being consumed
        f32x4 cm[NL][D][D];
        f32x4 cc[NL];
        const int rloc = ci < kpt ? ci : kpt - 1;              // padded lanes re-read the tile's last row
        const int swb = BF ? swz_h<KC>(rloc) : swz<KC>(rloc);
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            cc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < D; ++s2)
#pragma unroll
                for (int b = 0; b < D; ++b) cm[j][s2][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s <= D; ++s) {
            const uint32_t nid = (s < D) ? ids[s + 1] : ids_n[0];
            const float* nrow = row_of(nid);
            inv_keep[s] = inv_cur;
            inv_cur = a.inv[nid];
            if (s == D) issue_small(have_next ? nxt_tile : tile);
            // two buffers: slot s lives in buffer s & 1 (for even D the tile ends in buffer 0 again, for odd D
            // the next tile's slot 0 has landed in buffer 1 and is moved once per tile, below)
            const int bcur = (NB == 2) ? (s & 1) : 0;
            if constexpr (NB == 2) {
#pragma unroll
                for (int t = 0; t < KC; ++t) rb[bcur ^ 1][t] = load_chunk(nrow, t);
            }
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                const f32x4 cur = mask_chunk(rb[bcur][t], t);
                const int c = 4 * t + kq;
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const int jl = wpart + j * ics;                    // resident column tile of this wave
                    if (jl < nloc) {
                        const float* brow = bank + (size_t)(jl * kpt + rloc) * FP;
                        if (s < D) {
                            const int si = s < D ? s : 0;
                            if constexpr (BF) {
                                const short* hrow = (const short*)bank + (size_t)(jl * kpt + rloc) * FP;
                                s16x4 bh[D];
#pragma unroll
                                for (int b = 0; b < D; ++b) bh[b] = *(const s16x4*)(hrow + (size_t)b * SROW * FP + 4 * (c ^ swb));
                                const s16x4 a4 = to_bf16x4(cur);
#pragma unroll
                                for (int b = 0; b < D; ++b)
                                    cm[j][si][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, bh[b], cm[j][si][b], 0, 0, 0);
                            } else {
                            f32x4 bf[D];
#pragma unroll
                            for (int b = 0; b < D; ++b) bf[b] = *(const f32x4*)(brow + (size_t)b * SROW * FP + 4 * (c ^ swb));
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                                    for (int b = 0; b < D; ++b)
                                        cm[j][si][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q4], bf[b][q4], cm[j][si][b], 0, 0, 0);
                            }
                        } else {
                            if constexpr (BF) {
                                const short* hrow = (const short*)bank + (size_t)(jl * kpt + rloc) * FP;
                                const s16x4 bc = *(const s16x4*)(hrow + (size_t)D * SROW * FP + 4 * (c ^ swb));
                                cc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(to_bf16x4(cur), bc, cc[j], 0, 0, 0);
                            } else {
                            const f32x4 bc = *(const f32x4*)(brow + (size_t)D * SROW * FP + 4 * (c ^ swb));
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4)
                                    cc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q4], bc[q4], cc[j], 0, 0, 0);
                            }
                        }
                    }
                }
                if constexpr (NB == 1) rb[0][t] = load_chunk(nrow, t);  // the same chunk of the following slot
            }
        }
        MKGNN_STAMP(stamp_slot + 1);
        // row-owned values -> wave scratch (lanes kq == 0 own rows 0..15), read back 4 atoms at a time
        if (kq == 0) {
#pragma unroll
            for (int s = 0; s <= D; ++s) wscr[s * 16 + ci] = inv_keep[s];
            wscr[5 * 16 + ci] = __uint_as_float(focal_row);
            if constexpr (D == 4) {
                wscr[6 * 16 + ci] = sign_row;
                wscr[7 * 16 + ci] = __int_as_float(eq_row);
            }
        }
        f32x4 inv4[D + 1];
#pragma unroll
        for (int s = 0; s <= D; ++s) inv4[s] = *(const f32x4*)(wscr + s * 16 + kq * 4);
        const f32x4 focal4 = *(const f32x4*)(wscr + 5 * 16 + kq * 4);
        f32x4 sign4 = {0.f, 0.f, 0.f, 0.f}, eq4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (D == 4) {
            sign4 = *(const f32x4*)(wscr + 6 * 16 + kq * 4);
            eq4 = *(const f32x4*)(wscr + 7 * 16 + kq * 4);
        }

        // ---- epilogue per column tile: lane = kernel lcol, atoms kq*4 + jj
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int jl = wpart + j * ics;
            const int ct = cp + jl * cs;
            if (jl < nloc && ct < nct) {
                const int lcol = ct * kpt + ci;
                const bool col_ok = (ci < kpt) && (lcol < L);
                int idx4[4];
                float best4[4], cen4[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float m[D][D];
#pragma unroll
                    for (int s = 0; s < D; ++s) {
                        const float iv = inv4[s][jj];
#pragma unroll
                        for (int b = 0; b < D; ++b) m[s][b] = cm[j][s][b][jj] * iv;
                    }
                    best_permutation<D>(m, best4[jj], idx4[jj]);
                    cen4[jj] = cc[j][jj] * inv4[D][jj];
                }
#ifdef MKGNN_FWD_STAMPS
                if (stamps && stamp_slot == 2 && j == 0) { asm volatile("" :: "v"(best4[0]), "v"(best4[3])); MKGNN_STAMP(20); }
#endif
                // bond-cosine matrices, one (a, b) tile at a time; keep the entry the chosen order uses
                float ed4[4][D];
                {
                    const float* eb = esup + (size_t)(jl * kpt + rloc) * 8 + 2 * kq;
                    float2 bv[D];
#pragma unroll
                    for (int b = 0; b < D; ++b) bv[b] = *(const float2*)(eb + (size_t)b * SROW * 8);
#pragma unroll
                    for (int s = 0; s < D; ++s) {
                        f32x4 dm[D];              // D independent chains: the matrix pipe stays busy
#pragma unroll
                        for (int b = 0; b < D; ++b)
                            dm[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(eu[s].x, bv[b].x, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < D; ++b) dm[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(eu[s].y, bv[b].y, dm[b], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < D; ++b)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                if (b == 0 || perm_at<D>(idx4[jj], s) == b) ed4[jj][s] = dm[b][jj];
                    }
                }
#ifdef MKGNN_FWD_STAMPS
                if (stamps && stamp_slot == 2 && j == 0) { asm volatile("" :: "v"(ed4[0][0]), "v"(ed4[3][D - 1])); MKGNN_STAMP(21); }
#endif
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
                        if (do_chir && !__float_as_int(eq4[jj]))
                            ch = ((float)chirtab[(col_ok ? lcol : 0) * 12 + idx4[jj]] == sign4[jj]) ? 1.f : -1.f;
                        sc *= ch;
                    }
                    const uint32_t focal = __float_as_uint(focal4[jj]);
                    if (col_ok && n < dg.n) {
                        a.out[(size_t)focal * a.os + dg.off + lcol] = sc;
                        const uint32_t o = (uint32_t)n * (uint32_t)L + (uint32_t)lcol;   // the host fuses a degree only if N_d * L < 2^32
                        if (dg.pair) pair_store(dg.pair, o, best4[jj], cen4[jj], ed, idx4[jj]);
                        if (dg.chir_out) dg.chir_out[o] = (int8_t)ch;
                    }
                }
            }
        }
        MKGNN_STAMP(stamp_slot + 2);
        if (stamp_slot + 3 >= 20) stamp_slot -= 3;      // keep the fine stamps of the first tile
        stamp_slot += 3;
        if (!have_next) break;
        tile = nxt_tile;
#pragma unroll
        for (int s = 0; s <= D; ++s) ids[s] = ids_n[s];
        if constexpr (NB == 2 && (D & 1) == 0) {   // even D: D + 1 slots, the new slot 0 was fetched into buffer 1
#pragma unroll
            for (int t = 0; t < KC; ++t) rb[0][t] = rb[1][t];
        }
    }
}

template <int KC, bool GEN, bool BF>
__global__ void __launch_bounds__(256, 2) kc_forward_fused(FusedFwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int grp = a.blk_group[blockIdx.x];
    const int rank = a.blk_rank[blockIdx.x];
    const int di = a.grp_degree[grp];
    const int cp = a.grp_cp[grp];
    const int count = a.grp_count[grp];
    switch (di) {
        case 0: forward_body<1, KC, GEN, BF>(a, a.deg[0], cp, rank, count, lds); break;
        case 1: forward_body<2, KC, GEN, BF>(a, a.deg[1], cp, rank, count, lds); break;
        case 2: forward_body<3, KC, GEN, BF>(a, a.deg[2], cp, rank, count, lds); break;
        default: forward_body<4, KC, GEN, BF>(a, a.deg[3], cp, rank, count, lds); break;
    }
}

// eq[n] = 1 if any two of the four neighbour rows of degree-4 atom n are bit-identical
// (kernels.py:310-317: such an atom is not chiral).  One wave per atom.
__global__ void __launch_bounds__(256) rows_equal_kernel(const float* __restrict__ x, int64_t xs, const int64_t* __restrict__ nei,
                                                         const float* __restrict__ p_focal, const float* __restrict__ p_nei,
                                                         int64_t n, int F, int8_t* __restrict__ eq, int8_t* __restrict__ sgn, int x_split) {
    const int lane = threadIdx.x & 63;
    // pre-split rows (kgnn_split.h): a row's F floats are the same F (rounded up to 4: zero padding) dwords, each two fp16
    // halves of the scaled row; two rows of equal norm are equal exactly when their halves are (+0 and -0 being equal values,
    // as under torch.equal) -- rows of different norms differ in some half, or are scaled copies of each other by a power of
    // two, which h = propagate(sim_sc) does not produce short of an all-zero row
    if (x_split) F = (F + 3) / 4 * 4;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n; r += nwaves) {
        int64_t nb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) nb[j] = nei[r * 4 + j];
        // the coordinates travel with the ids (they were one more dependent round trip at the end, under lane == 0)
        float pn[3][3], pf[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pf[c] = p_focal[r * 3 + c];
#pragma unroll
            for (int j = 0; j < 3; ++j) pn[j][c] = p_nei[(r * 4 + j) * 3 + c];
        }
        unsigned diff = 0;
        for (int f0 = 0; f0 < F; f0 += 128) {             // two column blocks per pass, all eight loads in flight
            const int fa = f0 + lane, fb = f0 + 64 + lane;
            float va[4], vb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[j] = x[nb[j] * xs + (fa < F ? fa : 0)];
                vb[j] = x[nb[j] * xs + (fb < F ? fb : 0)];
            }
            int k = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i + 1; j < 4; ++j, ++k) {
                    if (x_split) {
                        auto same = [](float p, float q) {
                            const uint32_t a = __float_as_uint(p), b = __float_as_uint(q);
                            const bool lo_ok = ((a ^ b) & 0xffffu) == 0 || ((a | b) & 0x7fffu) == 0;
                            const bool hi_ok = ((a ^ b) >> 16) == 0 || ((a | b) & 0x7fff0000u) == 0;
                            return lo_ok && hi_ok;
                        };
                        if (fa < F && !same(va[i], va[j])) diff |= 1u << k;
                        if (fb < F && !same(vb[i], vb[j])) diff |= 1u << k;
                    } else {
                        if (fa < F && !(va[i] == va[j])) diff |= 1u << k;
                        if (fb < F && !(vb[i] == vb[j])) diff |= 1u << k;
                    }
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor((int)diff, o, 64);
        if (lane == 0) {
            eq[r] = (diff != 0x3Fu) ? 1 : 0;
            // sign of the neighbour tetrahedron, coordinates relative to the focal atom (kernels.py:356, 327-337)
            float t3[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) t3[j][c] = __fsub_rn(pn[j][c], pf[c]);
            sgn[r] = (int8_t)triple_sign(t3[0], t3[1], t3[2]);
        }
    }
}

// ---------------------------------------------------------------- host ----
extern "C" int mkgnn_debug_set_stamp_buffer(void* device_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buffer), &device_ptr, sizeof(void*));
}

// Measurement hook for bench.py: when enabled, every fused forward launch is bracketed by HIP
// events on its own stream; the getter waits for the last one and returns its duration.
// (a measurement hook for one benchmarking thread: the flag is atomic, the event pair is not per caller)
// enable = R > 1: the forward kernel is launched R times back to back between ONE event pair (it is idempotent: the same
// output and pair records every time) and the getter returns the mean -- an event pair around a single launch adds its own
// few microseconds, which is 5-10 % of this kernel.
static std::atomic<bool> g_time_fused{false};
static std::atomic<int> g_time_reps{1};
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
extern "C" int mkgnn_debug_time_fused_forward(int enable) {
    if (enable && !g_ev0) {
        if (hipEventCreate(&g_ev0) != hipSuccess || hipEventCreate(&g_ev1) != hipSuccess) return 1;
    }
    g_time_reps.store(enable > 1 ? (enable > 1000 ? 1000 : enable) : 1);
    g_time_fused.store(enable != 0);
    return 0;
}
extern "C" float mkgnn_debug_last_fused_forward_ms(void) {
    float ms = -1.f;
    if (!g_ev0 || hipEventSynchronize(g_ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, g_ev0, g_ev1) != hipSuccess) return -1.f;
    return ms / (float)g_time_reps.load();
}

bool mfma_forward_supported(int d, int F, int E, int L) {
    if (d < 1 || d > 4 || L < 1 || E > 8) return false;
    const int FP = mfma_padded_width(F);
    if (!FP) return false;
    const int nct = (L + 15) / 16;
    const int nl = d >= 3 ? 1 : 2;
    const int need_split = (nct + nl - 1) / nl;
    if (need_split > 16) return false;                      // <= 4 waves x 4 column parts
    const int kpt = (L + nct - 1) / nct;
    return (size_t)fused_lds_floats(d, FP / 16, L, 1, kpt) * 4 <= 60 * 1024;
}

// (degree, column part) groups a degree occupies in the fused launch (the same splitting rule as plan_fused)
int fused_group_count(int d, int F, int L) {
    const int KC = mfma_padded_width(F) / 16;
    const int nct = (L + 15) / 16;
    const int kpt = (L + nct - 1) / nct;
    const int nl = d >= 3 ? 1 : 2;
    int cs = (nct + nl - 1) / nl;
    int nloc = (nct + cs - 1) / cs;
    while ((size_t)fused_lds_floats(d, KC, L, nloc, kpt) * 4 > 60 * 1024 && cs < nct) {
        ++cs;
        nloc = (nct + cs - 1) / cs;
    }
    return cs;
}

// Fill geometry, group sizes and the block table; returns the dynamic LDS bytes (0 = nothing to launch).
static size_t plan_fused(FusedFwdArgs& a, const bool use[4], int KC, int* nblocks_out) {
    double cost[FUSED_MAX_GROUPS];                   // per 16-atom tile
    int64_t cap[FUSED_MAX_GROUPS], tiles_of[FUSED_MAX_GROUPS];
    int ng = 0;
    size_t lds_floats = 0;
    for (int i = 0; i < 4; ++i) {
        FusedDeg& g = a.deg[i];
        if (!use[i]) continue;
        const int d = i + 1, L = g.L;
        g.nct = (L + 15) / 16;
        g.kpt = (L + g.nct - 1) / g.nct;            // balanced: 10, 10, 15, 13 kernels for L = 10, 20, 30, 50
        const int64_t ntiles = (g.n + 15) / 16;
        // Column tiles are spread (a) over the waves of a block that share an atom tile (ics; their
        // row loads coalesce in L1) and (b) over blocks (cs; each part gathers the rows again), so that
        // one wave accumulates at most NL column tiles and a block's bank share stays <= ~57 KB.
        const int nl = d >= 3 ? 1 : 2;
        const int need_split = (g.nct + nl - 1) / nl;       // waves/blocks a tile's columns must be spread over
        g.ics = 1;
        g.cs = need_split;
        g.nloc = (g.nct + g.cs - 1) / g.cs;
        // the bank share must leave room for two blocks per CU
        while ((size_t)fused_lds_floats(d, KC, L, g.nloc, g.kpt) * 4 > 60 * 1024 && g.cs < g.nct) {
            ++g.cs;
            g.nloc = (g.nct + g.cs - 1) / g.cs;
        }
        const size_t fl = (size_t)fused_lds_floats(d, KC, L, g.nloc, g.kpt);
        if (fl > lds_floats) lds_floats = fl;
        for (int cp = 0; cp < g.cs && ng < FUSED_MAX_GROUPS; ++cp) {
            int mine = 0;                            // column tiles of this part
            for (int ct = cp; ct < g.nct; ct += g.cs) ++mine;
            const int tiles_per_block = 4 / g.ics;
            a.grp_degree[ng] = (uint8_t)i;
            a.grp_cp[ng] = (uint8_t)cp;
            // MFMAs per tile of this part + a term for the gather (rows are fetched once per part)
            // (calibrated against cycle stamps at F = 110: ~100 / 80 / 75 / 97 cycles per unit for degree 1..4)
            static const double calib[4] = {1.25, 1.0, 0.95, 1.2};
            cost[ng] = calib[i] * (mine * ((d * d + 1) * 4.0 * KC + 2.0 * d * d) + 0.35 * (d + 1) * 4.0 * KC);
            tiles_of[ng] = ntiles;
            cap[ng] = (ntiles + tiles_per_block - 1) / tiles_per_block;
            ++ng;
        }
    }
    if (ng == 0) { *nblocks_out = 0; return 0; }
    // Block counts: all blocks are resident at once (two per CU) and a wave walks its group's tiles round
    // robin, so the launch lasts as long as its slowest wave: prologue + ceil(tiles / waves) * cost per tile.
    // A share proportional to the total cost ignores that ceil(): with 2-5 tiles per wave at batch 4096 the
    // degree-4 groups got 2.4 tiles per wave on average, 3 on the slowest, and set the kernel time (stamps:
    // 170 k cycles against 135 k for the other groups).  Greedy min-max instead: every block goes to the
    // group that currently finishes last.
    const double prologue = 100.0;                   // bank copy + first gather, in the cost unit (~80 cycles)
    auto finish = [&](int g, int blocks) {
        const int64_t waves = 4 * (int64_t)blocks;
        return prologue + (double)((tiles_of[g] + waves - 1) / waves) * cost[g];
    };
    int count[FUSED_MAX_GROUPS];
    int nb = 0;
    for (int g = 0; g < ng; ++g) { count[g] = 1; ++nb; }
    while (nb < FUSED_MAX_BLOCKS) {
        int worst = -1;
        double t_worst = -1.0;
        for (int g = 0; g < ng; ++g) {
            if (count[g] >= cap[g]) continue;
            const double t = finish(g, count[g]);
            if (t > t_worst) { t_worst = t; worst = g; }
        }
        if (worst < 0) break;
        // blocks are only useful up to the next drop of ceil(tiles / waves)
        ++count[worst]; ++nb;
    }
    // interleave the groups over the block ids (largest remaining share first), so that
    // consecutive blocks -- which the dispatcher deals round-robin over the XCDs -- mix degrees
    int given[FUSED_MAX_GROUPS] = {0};
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
    // ranks: the blocks of a group that land on the same XCD (block id mod 8, round-robin dispatch) get consecutive
    // ranks, i.e. adjacent tile runs
    {
        int per_xcd[FUSED_MAX_GROUPS][8] = {};
        for (int b = 0; b < nb; ++b) ++per_xcd[a.blk_group[b]][b & 7];
        int next[FUSED_MAX_GROUPS][8];
        for (int g = 0; g < ng; ++g) {
            int run = 0;
            for (int x = 0; x < 8; ++x) { next[g][x] = run; run += per_xcd[g][x]; }
        }
        for (int b = 0; b < nb; ++b) a.blk_rank[b] = (uint16_t)next[a.blk_group[b]][b & 7]++;
    }
    for (int g = 0; g < ng; ++g) a.grp_count[g] = (uint16_t)count[g];
    *nblocks_out = nb;
    return lds_floats * 4;
}

hipError_t launch_forward_fused(FusedFwdArgs& a, const bool use[4], hipStream_t st) {
    if (a.last && use[3] && a.deg[3].n > 0) {
        int64_t blocks = (a.deg[3].n + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        rows_equal_kernel<<<(int)blocks, 256, 0, st>>>(a.x, a.xs, a.deg[3].nei, a.deg[3].p_focal, a.deg[3].p_nei, a.deg[3].n, a.F,
                                                      (int8_t*)a.deg[3].eqflag, (int8_t*)a.deg[3].signflag, a.x_split);
    }
    // The reference's bank shapes take the streamed kernel (kgnn_fwd_stream.hip); every other covered shape, and the
    // bf16 variant, the LDS-bank kernel below.  MKGNN_FWD_STREAM=0: A/B switch (diagnostics).
    static const char* env_stream = getenv("MKGNN_FWD_STREAM");
    // (round 4: the bf16 variant too, for the model's row widths; MKGNN_BF16_STREAM=0 keeps it on the LDS-bank kernel)
    static const char* env_bf = getenv("MKGNN_BF16_STREAM");
    const bool bf_stream = !(env_bf && env_bf[0] == '0') && stream_forward_bf16_supported(a.F);
    const bool stream_on = !(env_stream && env_stream[0] == '0') && (!a.bf16 || bf_stream);
    bool use_stream[4], use_bank[4];
    bool any_bank = false;
    for (int i = 0; i < 4; ++i) {
        use_stream[i] = use[i] && stream_on && stream_forward_supported(i + 1, a.F, a.E, a.deg[i].L, a.n_atoms, a.xs, a.os, a.deg[i].e_unit);
        use_bank[i] = use[i] && !use_stream[i];
        any_bank = any_bank || use_bank[i];
    }
    const int KC = mfma_padded_width(a.F) / 16;          // (0: rows wider than the LDS-bank kernel takes -- streamed or nothing)
    {   // the streamed launch holds FUSED_MAX_GROUPS (degree, column part) groups: very wide banks go to the LDS-bank kernel, widest first
        int Ls[4];
        for (int i = 0; i < 4; ++i) Ls[i] = a.deg[i].L;
        while (stream_forward_groups(Ls, use_stream) > FUSED_MAX_GROUPS) {
            int big = -1;                                    // (only a degree the LDS-bank kernel covers can move there; the caller
            for (int i = 0; i < 4; ++i)                      //  keeps the stream-only degrees within the table by itself)
                if (use_stream[i] && mfma_forward_supported(i + 1, a.F, a.E, Ls[i]) &&
                    (big < 0 || stream_column_parts(i + 1, Ls[i]) > stream_column_parts(big + 1, Ls[big]))) big = i;
            if (big < 0) return hipErrorInvalidValue;
            use_stream[big] = false; use_bank[big] = true; any_bank = true;
        }
    }
    const int time_reps = g_time_fused.load() ? g_time_reps.load() : 1;
    if (g_time_fused.load()) (void)hipEventRecord(g_ev0, st);
    for (int rep = 0; rep < time_reps; ++rep) {
    {
        FusedFwdArgs s = a;
        hipError_t e = launch_forward_stream(s, use_stream, st);
        if (e != hipSuccess) return e;
    }
    if (any_bank) {
        if (KC == 0 || a.x_split) return hipErrorInvalidValue;      // (pre-split rows: the streamed kernel only; the C ABI checks first)
        int nb = 0;
        const size_t lds_bytes = plan_fused(a, use_bank, KC, &nb);
        if (nb > 0) {
            const bool gen = a.F <= 16 * (KC - 1);           // more than the last chunk can be partial or empty
            if (a.bf16) {
                if (KC == 2) { if (gen) kc_forward_fused<2, true, true><<<nb, 256, lds_bytes, st>>>(a); else kc_forward_fused<2, false, true><<<nb, 256, lds_bytes, st>>>(a); }
                else { if (gen) kc_forward_fused<7, true, true><<<nb, 256, lds_bytes, st>>>(a); else kc_forward_fused<7, false, true><<<nb, 256, lds_bytes, st>>>(a); }
            } else {
                if (KC == 2) { if (gen) kc_forward_fused<2, true, false><<<nb, 256, lds_bytes, st>>>(a); else kc_forward_fused<2, false, false><<<nb, 256, lds_bytes, st>>>(a); }
                else { if (gen) kc_forward_fused<7, true, false><<<nb, 256, lds_bytes, st>>>(a); else kc_forward_fused<7, false, false><<<nb, 256, lds_bytes, st>>>(a); }
            }
        }
    }
    }   // (timing repetitions: both launches of a call inside the event pair)
    if (g_time_fused.load()) (void)hipEventRecord(g_ev1, st);
    return hipGetLastError();
}

}  // namespace mkgnn
