// Streamed MFMA bank-gradient kernel of the kernel convolution (gfx950): the gradient with respect to the
// unit-normalised kernel rows (autograd of reference kernels.py:353-425 towards x_center / x_support /
// edge_attr_support), all four degree buckets in ONE launch, on the matrix pipe.
//
//   g_unit[l, b, :] = sum_n coef[n, l] * [pi_{n,l}(a) = b] * xhat[nei(n, a), :]          (supports; a = neighbour slot)
//   g_unit[l, c, :] = sum_n coefc[n, l] * xhat[n, :]                                      (centres)
//   g_edge[l, b, :] = sum_n coefe[n, l] * [pi_{n,l}(a) = b] * ehat[n, a, :]               (bond supports)
//
// Per 16-atom tile and 16-kernel column tile this is P_ab^T [kernels x atoms] . X_a [atoms x F]: the transpose of the
// forward's product with the 0/1-masked coefficient tile as the A operand.  It runs on the forward's streaming skeleton
// (kgnn_fwd_stream.hip): a wave owns a column tile and keeps ITS accumulators -- the gradient rows of its kernels, 28
// VGPRs per bank slot -- in registers for the whole launch; the gathered atom rows stream through an LDS ring filled by
// LDS-DMA (here in plain row-major order: a DMA piece is 64 consecutive 16-byte chunks of the 16 x 448-byte slot image,
// whole rows in full cache lines, and the B-operand reads -- one float per lane, atom k, feature j -- are conflict-free
// in that layout); counted vmcnt waits and raw barriers as in the forward.  There is no per-tile epilogue: a wave writes
// its accumulators once, at the end, as its slice of a partial slab, and kc_backward_bank_reduce sums the slabs in a
// fixed order (bit-reproducible; no float atomics).
//
// The per-(atom, kernel) inputs -- dL/dsc through the focal ids, the chosen permutation, the chirality sign -- are put
// into tile order by a small pre-pass (coef_prepare_kernel: one 1 KB + 1 KB record per (atom tile, column tile), so the
// main kernel fetches them with two DMA pieces per tile); the pre-pass also sums the three score-weight partials
// d sc / d theta_k = w_k (score_k - sc) / W, which need every pair's three scores exactly once.  It runs first, on the
// caller's stream (plan_backward_bank_stream / launch_coef_prepare / launch_backward_bank_stream): the rows kernel
// (kgnn_bwd_rows_stream.hip) reads the same records.  With BankStreamArgs.through_nei the caller's grad_out is the
// gradient of h = propagate(out) and the pre-pass sums the rows of an atom's neighbours (the bucket's nei_index: the
// targets of its edges, KernelLayer.py:119-123) where it otherwise reads the focal atom's row.
//
// Degree 4 splits a column tile's supports over two waves like the forward (its accumulators would not fit otherwise);
// covered shapes are the forward's (stream_forward_supported): any F <= 112 (KC = 1 .. 7 chunks), any number of kernels
// (column parts: a degree with more column tiles than a stream's waves hold is cut into parts, each a group of blocks; all
// parts of a degree get the same number of blocks, so part p's stream s fills its kernels' rows of slab chunk s).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "kgnn_launch.h"
#include "kgnn_split.h"

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build (make STAMPS=1): per-wave cycle totals of the loop's phases in a buffer set by
// mkgnn_debug_set_bwd_stream_stamps (tools/bwd_stream_stamps.py): [start, end, degree * 16 + column part, iterations,
// phase 0..7].  Bank kernel phases: 0 tile coefficients, 1 operand preparation (norms, bond rows, masks, bond products),
// 2 row products (B reads + matrix instructions), 3 counted wait, 4 barrier, 5 DMA issue, 6 slab store.
#ifdef MKGNN_BWD_STAMPS
__device__ unsigned long long* g_bank_stream_stamps = nullptr;
#define MKGNN_BPHASE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); phase[i] += t_ - t_phase; t_phase = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MKGNN_BPHASE(i) do { } while (0)
#endif

namespace bs {

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (B < E) { fn(IC<B>{}); static_for<B + 1, E>(fn); }
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "s_waitcnt vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// (SITE: one instruction offset per place that issues an LDS-DMA, so that the compiler cannot merge two of them into one
// instruction with a phi'd -- then readfirstlane'd -- LDS base: see kgnn_fwd_stream.hip)
#define MKGNN_DMA_CASE(SZ, O) else if constexpr (OFF == O) __builtin_amdgcn_global_load_lds(g, l, SZ, O, 0)
template <int OFF> __device__ __forceinline__ void dma16(const void* src, float* lds_wave_base) {
    const auto g = (const __attribute__((address_space(1))) void*)((const char*)src - OFF);
    const auto l = (__attribute__((address_space(3))) void*)((char*)lds_wave_base - OFF);
    if constexpr (OFF == 0) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    MKGNN_DMA_CASE(16, 16); MKGNN_DMA_CASE(16, 32); MKGNN_DMA_CASE(16, 48);
    else static_assert(OFF < 0, "add the offset to the list");
}
template <int OFF> __device__ __forceinline__ void dma4(const void* src, float* lds_wave_base) {
    const auto g = (const __attribute__((address_space(1))) void*)((const char*)src - OFF);
    const auto l = (__attribute__((address_space(3))) void*)((char*)lds_wave_base - OFF);
    if constexpr (OFF == 0) __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
    MKGNN_DMA_CASE(4, 4); MKGNN_DMA_CASE(4, 8); MKGNN_DMA_CASE(4, 12);
    else static_assert(OFF < 0, "add the offset to the list");
}
constexpr int SITE_ROWS = 0, SITE_BONDS = 16, SITE_COEF_G = 32, SITE_COEF_I = 48;  // dma16
constexpr int SITE_IDS = 0, SITE_IDS4 = 4, SITE_INV = 8, SITE_INV4 = 12;            // dma4
// an LDS dword written by DMA, read behind the compiler's back (it would drain the DMA queue in front of an ordinary
// read that may alias a pending DMA); valid after lds_fence over the same registers
__device__ __forceinline__ float lds_read_raw(uint32_t byte_addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr) : "memory");
    return v;
}
// ... with a compile-time byte offset in the instruction (one address register for a whole family of reads)
template <int OFF> __device__ __forceinline__ float lds_read_raw_at(uint32_t byte_addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF) : "memory");
    return v;
}

// a 16-bit LDS read, zero-extended.  (Not ds_read_u16_d16 / _d16_hi into the two halves of ONE register: with SRAM ECC on --
// every MI300 / MI355 -- a d16 load rewrites the whole register, the "kept" half comes back as zero; found by
// tests/test_hip_parity.py::test_three_layer_network_tie_aware, layer 1's bank gradients.  Two reads and one v_lshl_or_b32.)
template <int OFF> __device__ __forceinline__ void lds_read_h16(uint32_t& dst, uint32_t byte_addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(byte_addr), "n"(OFF) : "memory");
}

template <typename V> __device__ __forceinline__ void lds_fence(V& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }

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

template <int D> struct Traits {
    static constexpr int NS = (D == 1) ? 1 : (D == 4 ? 4 : 2);     // waves sharing an atom tile
    static constexpr int NSTREAM = 4 / NS;
    static constexpr int RING = (D == 1) ? 2 : (D == 2 ? 3 : (D == 3 ? 4 : 5));   // = the forward's (<= S1)
    static constexpr int S1 = D + 1;
    // per-tile record (floats): ids[S1][16] | inv[S1][16] | bond[D][16][8]
    static constexpr int OFF_INV = 16 * S1, OFF_BOND = 32 * S1, META = OFF_BOND + 128 * D;
    static constexpr int COEF = 512;                               // per wave and tile: g[16][16] floats | idx[16][16] ints
};

__host__ __device__ constexpr int lds_floats(int D, int KC) {
    const int NS = (D == 1) ? 1 : (D == 4 ? 4 : 2);
    const int RING = (D == 1) ? 2 : (D == 2 ? 3 : (D == 3 ? 4 : 5));
    const int META = 32 * (D + 1) + 128 * D;
    const int FPC = 4 * KC;
    return (4 / NS) * (RING * 16 * FPC * 4 + 2 * META) + 4 * 512;
}

template <int KC, int NS> constexpr int pieces_of(int role) { return (KC - role + NS - 1) / NS; }

// vector-memory operations wave `role` issues in the DMA phase whose slot is `sd`
template <int D, int KC> constexpr int batch_size(int sd, int role) {
    using T = Traits<D>;
    int n = pieces_of<KC, T::NS>(role);
    if (sd == 0) n += 2;                                 // this wave's coefficient record of the tile (g, idx)
    if (role == 0) {
        if (sd < D) n += 1;                              // unit bond rows of the slot
        if (sd == 0) n += 2 * (T::S1 == 5 ? 2 : 1);      // following tile's ids, this tile's 1/|x|
    }
    return n;
}
template <int D, int KC> constexpr int young_batches(int s, int role) {
    using T = Traits<D>;
    int n = 0;
    for (int j = 1; j <= T::RING - 2; ++j) n += batch_size<D, KC>((s + 64 * T::S1 - j + T::RING) % T::S1, role);
    return n;
}

}  // namespace bs

// ------------------------------------------------------------- pre-pass ---
// One thread per (atom of a tile, column of a column tile): the pair's dL/dsc (times the chirality sign) and
// permutation id into tile order, zeros for padding; the three score-weight partials summed per block in a fixed order.
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
#ifndef MKGNN_PREP_RPB
#define MKGNN_PREP_RPB 4
#endif
constexpr int PREP_RPB = MKGNN_PREP_RPB;                              // records per block: four independent load chains per thread
// NSRC: rows of grad_out a pair's dL/dsc is summed from -- 1 (the focal atom's) or, through_nei, the degree's D neighbours'.
// A template parameter since round 5: with a run-time count the four clamped loads of ids and of rows were issued for every
// degree (the kernel is bound by its vector-memory instructions: TA busy 60 %), two of three of them redundant at degree 1 and
// half of them at degree 2, the largest group.
// CH: the degree keeps chirality signs (degree 4 of the last layer) -- otherwise their load is not issued at all.
template <int NSRC, bool CH>
__device__ __forceinline__ void coef_prepare_body(const BankStreamArgs& a, const int di) {
    const BankStreamDeg& g = a.deg[di];
    const int tid = threadIdx.x;
    const int64_t blk = (int64_t)blockIdx.x - g.prep_blk0;
    const int64_t nrec = ((g.n + 15) / 16) * g.nct;
    const int atom = tid >> 4, k = tid & 15;
    const float w_s = g.mix[0], w_c = g.mix[1], w_e = g.mix[2], w_sum = g.mix[3];
    const int8_t* chp = g.chir ? g.chir : (const int8_t*)g.pair;      // always loadable
    // all loads of the block's records first (clamped addresses), then the arithmetic
    float gv[PREP_RPB], S[PREP_RPB], C[PREP_RPB], Ed[PREP_RPB];
    int idx[PREP_RPB], ch[PREP_RPB];
    bool ok[PREP_RPB];
    // the rows of grad_out this pair's dL/dsc comes from: the focal atom's, or (through_nei: the propagate step's
    // gradient folded in, KernelLayer.py:119-123) the rows of its D neighbours, summed in slot order
    const int D = di + 1;
    constexpr int nsrc = NSRC;
    int64_t src[PREP_RPB][NSRC];
#pragma unroll
    for (int r = 0; r < PREP_RPB; ++r) {
        const int64_t rec = blk * PREP_RPB + r;
        const int64_t rc = rec < nrec ? rec : nrec - 1;
        const int64_t tile = rc / g.nct;
        const int ct = (int)(rc - tile * g.nct);
        const int64_t n = tile * 16 + atom;
        const int64_t nc = n < g.n ? n : g.n - 1;
        const int l = ct * g.kpt + k;
        ok[r] = rec < nrec && n < g.n && k < g.kpt && l < g.L;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) src[r][s] = a.through_nei ? g.nei[nc * D + s] : g.sel[nc];
    }
#pragma unroll
    for (int r = 0; r < PREP_RPB; ++r) {
        const int64_t rec = blk * PREP_RPB + r;
        const int64_t rc = rec < nrec ? rec : nrec - 1;
        const int64_t tile = rc / g.nct;
        const int ct = (int)(rc - tile * g.nct);
        const int64_t n = tile * 16 + atom;
        const int64_t nc = n < g.n ? n : g.n - 1;
        const int l = ct * g.kpt + k, lc = l < g.L ? l : g.L - 1;
        const size_t o = (size_t)nc * g.L + lc;
        float gs4[NSRC];
#pragma unroll
        for (int s = 0; s < NSRC; ++s) gs4[s] = a.gout[src[r][s] * a.gs + g.off + lc];
        gv[r] = gs4[0];
#pragma unroll
        for (int s = 1; s < NSRC; ++s) gv[r] += gs4[s];                  // (slot order, as before)
        const mkgnn_f32x4 pr = pair_load(g.pair, o);
        idx[r] = pair_index(pr);
        ch[r] = CH ? (int)chp[o] : 1;
        S[r] = pr[0]; C[r] = pr[1]; Ed[r] = pr[2];
    }
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    constexpr int NEM = (PREP_RPB + 1) / 2;
    pk_u16 em[NEM];                                      // biased exponents of |g| of the block's records, two per register
#pragma unroll
    for (int i = 0; i < NEM; ++i) em[i] = pk_u16{0, 0};
#pragma unroll
    for (int r = 0; r < PREP_RPB; ++r) {
        const int64_t rec = blk * PREP_RPB + r;
        float gg = ok[r] ? (CH ? gv[r] * (float)ch[r] : gv[r]) : 0.f;
        if (rec < nrec) {
            float* out = g.coefq + (size_t)rec * 512;
            out[tid] = gg;
        }
        const unsigned short eb = (unsigned short)((__float_as_uint(gg) >> 23) & 0xffu);
        em[r >> 1][r & 1] = eb;
        if (ok[r]) {
            const float sc = (S[r] * w_s + C[r] * w_c + Ed[r] * w_e) / w_sum;
            p0 = fmaf(gg * (w_s / w_sum), S[r] - sc, p0);
            p1 = fmaf(gg * (w_c / w_sum), C[r] - sc, p1);
            p2 = fmaf(gg * (w_e / w_sum), Ed[r] - sc, p2);
        }
    }
    // fixed-order block sums; the largest exponent of every KERNEL COLUMN of the records (a maximum: order-free).  Per column since
    // round 6: with one exponent per record a column whose own dL/dsc are 2^-24 of a sibling's lost the lo halves
    // of its coefficients in the bank kernel's split-fp16 products.  A thread is (atom tid >> 4, column tid & 15): the maximum
    // over the four atoms of its wave by two xor steps, over the four waves through LDS.
    __shared__ float red[3][4];
    __shared__ uint32_t emx[NEM][4][16];
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1)
#pragma unroll
        for (int i = 0; i < NEM; ++i)
            em[i] = __builtin_elementwise_max(em[i], __builtin_bit_cast(pk_u16, __shfl_xor(__builtin_bit_cast(uint32_t, em[i]), o, 64)));
    if ((tid & 63) == 0) { red[0][tid >> 6] = p0; red[1][tid >> 6] = p1; red[2][tid >> 6] = p2; }
    if ((tid & 63) < 16) {
#pragma unroll
        for (int i = 0; i < NEM; ++i) emx[i][tid >> 6][tid & 15] = __builtin_bit_cast(uint32_t, em[i]);
    }
    __syncthreads();
    {
        // idx words: the permutation id in the low byte, the largest exponent of the word's own kernel column above it (what the
        // bank kernel's split-fp16 products scale by: kgnn_split.h) -- every word of a column carries it, no reduction where it is read
        pk_u16 mx[NEM];
#pragma unroll
        for (int i = 0; i < NEM; ++i) {
            mx[i] = __builtin_bit_cast(pk_u16, emx[i][0][k]);
#pragma unroll
            for (int w = 1; w < 4; ++w) mx[i] = __builtin_elementwise_max(mx[i], __builtin_bit_cast(pk_u16, emx[i][w][k]));
        }
#pragma unroll
        for (int r = 0; r < PREP_RPB; ++r) {
            const int64_t rec = blk * PREP_RPB + r;
            const int me = mx[r >> 1][r & 1];
            if (rec < nrec) ((int*)(g.coefq + (size_t)rec * 512))[256 + tid] = (ok[r] ? idx[r] : 0) | (me << 8);
        }
    }
    if (tid < 3) g.theta_slab[(size_t)blk * 4 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

__global__ void __launch_bounds__(256) coef_prepare_kernel(BankStreamArgs a) {
    int di = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (a.deg[k].prep_blocks > 0 && (int)blockIdx.x >= a.deg[k].prep_blk0) di = k;
    const int nsrc = a.through_nei ? di + 1 : 1;         // (block-uniform, like the sign test)
    if (a.deg[di].chir) {
        switch (nsrc) {
            case 1: coef_prepare_body<1, true>(a, di); break;
            case 2: coef_prepare_body<2, true>(a, di); break;
            case 3: coef_prepare_body<3, true>(a, di); break;
            default: coef_prepare_body<4, true>(a, di); break;
        }
        return;
    }
    switch (nsrc) {
        case 1: coef_prepare_body<1, false>(a, di); break;
        case 2: coef_prepare_body<2, false>(a, di); break;
        case 3: coef_prepare_body<3, false>(a, di); break;
        default: coef_prepare_body<4, false>(a, di); break;
    }
}

// ------------------------------------------------------------ main body ---
// SP = true (round 5): the row products as split fp16 (kgnn_split.h).  k-position (lane >> 4, i) of the 16-deep product
// stands for atom 4 i + (lane >> 4): the four values a lane holds today -- coefficients of kernel ci, feature ci of the rows,
// atoms 4 q + kq -- are its A and B operands as they are.  Scales: a row (B, atom k) by 2^(exponent(1 / |x_k|) + 8), its
// inverse moved into the coefficient of the same atom (A: g w mantissa(1 / |x_k|) 2^-8); the coefficients by ONE power of
// two per wave, G, chosen so that the largest coefficient seen so far (the pre-pass leaves every record's largest exponent in
// its idx words) sits below 2^19 before the mantissa factor -- when a tile
// brings a larger one, G drops and the accumulators (which live in registers for the whole launch) are multiplied by the
// ratio, exactly; the slab slice is divided by G at the end.  Bit-reproducible: G depends on the wave's own tiles in order.
// SP = 2 (round 6): the atom rows arrive pre-split (kgnn_split.h, split_row_store: hi(0..3) | lo(0..3) per four floats, scaled by
// the same 2^(exponent(1 / |x|) + 8)): the B operand's halves are read straight out of the slot image with 16-bit LDS loads
// (ds_read_u16: eight per feature tile instead of four 32-bit ones, paired by four v_lshl_or_b32) and the ten conversion
// instructions per tile go.
constexpr int BANK_COEF_EXP = 18;
template <int D, int KC, int SP>
__device__ __forceinline__ void bank_stream_body(const BankStreamArgs& a, const BankStreamDeg& dg, const int cp, const int rank,
                                                 const int count, float* lds) {
    using namespace bs;
    using T = Traits<D>;
    constexpr int NS = T::NS, NSTREAM = T::NSTREAM, RING = T::RING, S1 = T::S1, META = T::META;
    constexpr int FPC = 4 * KC;                          // 16-byte chunks per row in the slot image
    // rows of an even number of 16-float chunks are a multiple of 32 banks apart: the transposed reads below (lane = atom
    // row kq, feature ci) would collide pairwise, so chunk c of an odd row sits at c ^ 4 (its 16-float halves swapped)
    constexpr bool SWZ = (KC % 2 == 0);
    constexpr int RF = 4 * FPC;                          // floats per row (112 / 32)
    constexpr int SLOT = 16 * RF;                        // floats per slot image
    constexpr int NP = SLOT / 256;                       // DMA pieces per slot (7 / 2 = KC)
    static_assert(NP == KC, "piece count");
    constexpr bool HS = (D == 4);
    constexpr int NBS = HS ? 2 : D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stream = wave / NS, role = wave % NS;
    const int half = HS ? (role & 1) : 0;
    const int ct = HS ? cp * 2 + (role >> 1) : cp * NS + role;
    const bool ct_ok = ct < dg.nct;                      // (past the degree's last column tile: an idle wave, zeros throughout)
    const int ctc = ct_ok ? ct : dg.nct - 1;
    const int ci = lane & 15, kq = lane >> 4;
    const int L = dg.L, kpt = dg.kpt;
    float* const ring = lds + (size_t)stream * (RING * SLOT + 2 * META);
    float* const meta = ring + RING * SLOT;
    // (one record per wave is enough: it is read into registers at the top of a tile, the next tile's DMA is issued later)
    float* const cbuf = lds + (size_t)NSTREAM * (RING * SLOT + 2 * META) + (size_t)wave * T::COEF;

    const int64_t ntiles = (dg.n + 15) / 16;
    const int64_t nstreams = (int64_t)count * NSTREAM;
    const int64_t sg = (int64_t)rank * NSTREAM + stream;
    const int64_t tile_first = sg * ntiles / nstreams;
    const int64_t tile_end = (sg + 1) * ntiles / nstreams;
    const int64_t iters = (ntiles + nstreams - 1) / nstreams;
    const int64_t tile_hi = (tile_end > tile_first ? tile_end : (tile_first + 1 < ntiles ? tile_first + 1 : ntiles)) - 1;
    auto tile_at = [&](int64_t i) -> int64_t {
        const int64_t t = tile_first + i;
        return t > tile_hi ? tile_hi : t;
    };
    const float w_s = dg.mix[0], w_c = dg.mix[1], w_e = dg.mix[2], w_sum = dg.mix[3];
    const float ws_n = w_s / w_sum / (float)D, wc_n = w_c / w_sum, we_n = w_e / w_sum / (float)D;
    const uint32_t xs = (uint32_t)a.xs;
    const int F = a.F;

    // DMA piece t of a slot: lane q fetches chunk (64 t + q) of the row-major 16 x FPC image (SWZ: see above)
    auto issue_rows = [&](auto sdc, const float* drec, float* buf) {
        constexpr int sd = decltype(sdc)::value;
        const uint32_t ids_b = (uint32_t)(uintptr_t)(drec + 16 * sd);
        static_for<0, NS>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int NMINE = pieces_of<KC, NS>(r);
            if constexpr (NMINE > 0) if (role == r) {
                uint32_t id[NMINE];
                int cc[NMINE];
                static_for<0, NMINE>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int g = 64 * (r + NS * k) + lane;
                    const int row = g / FPC;
                    int c = g - row * FPC;
                    if constexpr (SWZ) c ^= (row & 1) * 4;
                    cc[k] = c;
                    id[k] = __float_as_uint(lds_read_raw(ids_b + 4u * row));
                });
                static_for<0, NMINE>([&](auto kc) { lds_fence(id[decltype(kc)::value]); });
                static_for<0, NMINE>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    dma16<SITE_ROWS>(a.x + (id[k] * xs + (4 * cc[k] < F ? 4u * cc[k] : 0u)), buf + (r + NS * k) * 256);
                });
            }
        });
    };
    auto issue_bonds = [&](int64_t t, int sd, float* mrec) {
        if (lane < 32) {
            int64_t n = t * 16 + (lane >> 1);
            if (n >= dg.n) n = dg.n - 1;
            dma16<SITE_BONDS>(dg.e_unit + ((uint32_t)(n * D + sd) * 8u + 4u * (lane & 1)), mrec + T::OFF_BOND + sd * 128);
        }
    };
    auto issue_ids = [&](int64_t t, float* mrec) {
        int64_t n = t * 16 + ci;
        if (n >= dg.n) n = dg.n - 1;
        const void* src = (kq < D) ? (const void*)(dg.nei + n * D + kq) : (const void*)(dg.sel + n);
        if (S1 >= 4 || kq < S1) dma4<SITE_IDS>(src, mrec);
        if constexpr (S1 == 5) {
            if (kq == 0) dma4<SITE_IDS4>(dg.sel + n, mrec + 64);
        }
    };
    auto issue_inv = [&](float* mrec) {                  // 1 / |x| of every slot, through the ids in the record
        const uint32_t rec_b = (uint32_t)(uintptr_t)mrec;
        uint32_t idq = __float_as_uint(lds_read_raw(rec_b + 4u * (16 * ((S1 >= 4 || kq < S1) ? kq : 0) + ci)));
        [[maybe_unused]] uint32_t id4 = 0;
        if constexpr (S1 == 5) id4 = __float_as_uint(lds_read_raw(rec_b + 4u * (64 + ci)));
        lds_fence(idq);
        if constexpr (S1 == 5) lds_fence(id4);
        if (S1 >= 4 || kq < S1) dma4<SITE_INV>(a.inv + idq, mrec + T::OFF_INV);
        if constexpr (S1 == 5) {
            if (kq == 0) dma4<SITE_INV4>(a.inv + id4, mrec + T::OFF_INV + 64);
        }
    };
    auto issue_coef = [&](int64_t t, float* cb) {        // this wave's (tile, column tile) record: two 1 KB pieces
        const float* src = dg.coefq + ((size_t)(t * dg.nct + ctc) * 512 + 4 * lane);
        dma16<SITE_COEF_G>(src, cb);
        dma16<SITE_COEF_I>(src + 256, cb + 256);
    };

    f32x4 acc[NBS][KC];                                  // gradient rows of this wave's kernels: support slots
    f32x4 accC[KC];                                      // ... centre rows (HS: half 0)
    f32x4 accE[NBS];                                     // ... bond supports
#pragma unroll
    for (int t = 0; t < KC; ++t) {
        accC[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NBS; ++b) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < NBS; ++b) accE[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (SP) the coefficients' power-of-two scale, one per KERNEL COLUMN since round 6: this lane's A operand is kernel ci's column,
    // G is that column's scale (the same in the four lanes kq of a column: they read the same records in the same order)
    [[maybe_unused]] float G = __uint_as_float(230u << 23);      // 2^103: no coefficient seen yet
    // (exponent of the larger weight, + 1 for the mantissas' product, re-biased: added to a record's largest exponent)
    [[maybe_unused]] const int wexp = (int)((__float_as_uint(fmaxf(fabsf(ws_n), fabsf(wc_n))) >> 23) & 0xffu) + 1 - 127;

    // ---- prologue
    if (role == 0) issue_ids(tile_at(0), meta);
    wait_vmcnt<0>();
    __syncthreads();
    static_for<0, RING>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        issue_rows(kc, meta, ring + k * SLOT);
        if (role == 0 && k < D) issue_bonds(tile_at(0), k, meta);
    });
    issue_coef(tile_at(0), cbuf);
    if (role == 0) {
        issue_inv(meta);
        issue_ids(tile_at(1), meta + META);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    int buf = 0;
#ifdef MKGNN_BWD_STAMPS
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_phase = t_start;
#endif
    for (int64_t it = 0; it < iters; ++it) {
        const bool real = tile_first + it < tile_end;    // a repeated tile must not be accumulated twice
        const uint32_t mrec_b = (uint32_t)(uintptr_t)(meta + (it & 1) * META);            // LDS byte addresses
        const uint32_t cb_b = (uint32_t)(uintptr_t)cbuf;
        // the tile's coefficients for this lane: kernel ci, atoms 4 q + kq
        float gq[4];
        int iq[4];
        [[maybe_unused]] int rec_exp = 0;
        {
            float raw[8];
            const uint32_t cl_b = cb_b + 4u * (kq * 16 + ci);
            static_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                raw[q] = lds_read_raw_at<4 * (4 * q * 16)>(cl_b);
                raw[4 + q] = lds_read_raw_at<4 * (256 + 4 * q * 16)>(cl_b);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]),
                         "+v"(raw[6]), "+v"(raw[7]) : : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gq[q] = (real && ct_ok) ? raw[q] : 0.f;
                iq[q] = __float_as_int(raw[4 + q]) & 0xff;
            }
            if constexpr (SP) rec_exp = (__float_as_int(raw[4]) >> 8) & 0xff;     // (every idx word of the record carries it)
        }
        if constexpr (SP) {
            // a column's scale: never larger than what this tile's largest coefficient of that column (times the weights) allows.
            // The accumulators of kernel 4 kq + r sit in register element r of the lanes kq (all features ci): when some
            // column's scale drops, every lane fetches the ratios of ITS four kernels from the lanes that hold them as columns
            // (lane index = kernel) and rescales, exactly (powers of two) -- a few times per column and launch
            const float need = split_scale_for_exponent<BANK_COEF_EXP>(rec_exp + wexp);
            const float ratio_col = need < G ? need / G : 1.f;
            if (__any(need < G)) {                       // (wave-uniform)
                float rr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[r] = __shfl(ratio_col, 4 * kq + r, 64);
#pragma unroll
                for (int t = 0; t < KC; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accC[t][r] *= rr[r];
#pragma unroll
                        for (int b = 0; b < NBS; ++b) acc[b][t][r] *= rr[r];
                    }
                G = need < G ? need : G;
            }
        }
        MKGNN_BPHASE(0);
        static_for<0, S1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s < D || !HS || half == 0) {
                const uint32_t rb_b = (uint32_t)(uintptr_t)(ring + buf * SLOT);
                // 1 / |x| of the slot's rows for this lane's four atoms, the A operands, then the slot's rows as B operands
                float iv[4];
                const uint32_t il_b = mrec_b + 4u * kq;
                static_for<0, 4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    iv[q] = lds_read_raw_at<4 * (T::OFF_INV + s * 16 + 4 * q)>(il_b);
                });
                [[maybe_unused]] float eb[4];
                if constexpr (s < D) {
                    const uint32_t el_b = mrec_b + 4u * (kq * 8 + (ci & 7));
                    static_for<0, 4>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        eb[q] = lds_read_raw_at<4 * (T::OFF_BOND + (s < D ? s : 0) * 128 + 4 * q * 8)>(el_b);
                    });
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(iv[0]), "+v"(iv[1]), "+v"(iv[2]), "+v"(iv[3]), "+v"(eb[0]), "+v"(eb[1]),
                                 "+v"(eb[2]), "+v"(eb[3]) : : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(iv[0]), "+v"(iv[1]), "+v"(iv[2]), "+v"(iv[3]) : : "memory");
                }
                float av[NBS][4], ae[NBS][4], ac[4];
                [[maybe_unused]] f32x4 rsc;               // (SP) the rows' scales, atoms 4 q + kq
                if constexpr (SP) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (SP == 1) rsc[q] = __uint_as_float((__float_as_uint(iv[q]) & 0x7f800000u) + (8u << 23));
                        iv[q] = __uint_as_float((__float_as_uint(iv[q]) & 0x007fffffu) | ((127u - 8u) << 23)) * G;     // mantissa 2^-8 G
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if constexpr (s < D) {
                        const float cf = gq[q] * ws_n * iv[q], ce = gq[q] * we_n;
                        const int pb = perm_entry<D, s>(iq[q]);
#pragma unroll
                        for (int b = 0; b < NBS; ++b) {
                            const bool hit = pb == (HS ? 2 * half + b : b);
                            av[b][q] = hit ? cf : 0.f;
                            ae[b][q] = hit ? ce : 0.f;
                        }
                    } else {
                        ac[q] = gq[q] * wc_n * iv[q];
                    }
                }
                [[maybe_unused]] SplitReg avs[NBS];       // (SP) the masked coefficient operands of the slot, split once
                if constexpr (SP) {
                    if constexpr (s < D) {
#pragma unroll
                        for (int b = 0; b < NBS; ++b) avs[b] = split_exact(f32x4{av[b][0], av[b][1], av[b][2], av[b][3]});
                    } else {
                        avs[0] = split_exact(f32x4{ac[0], ac[1], ac[2], ac[3]});
                    }
                }
                if constexpr (s < D) {
#pragma unroll
                    for (int b = 0; b < NBS; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) accE[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[b][q], eb[q], accE[b], 0, 0, 0);
                }
                MKGNN_BPHASE(1);
                // feature tiles: lane (k = kq, j = ci) reads row 4 q + kq, feature 16 t + ci of the slot image
                // (row 4 q + kq is odd exactly when kq is: SWZ moves its column 16 t + ci to 16 (t ^ 1) + ci, i.e. 16 floats up
                // for even t and down for odd t -- two per-lane bases, the offsets stay immediates)
                const uint32_t xl_b = rb_b + 4u * (kq * RF + ci + (SWZ ? (kq & 1) * 16 : 0));
                [[maybe_unused]] const uint32_t xl_o = rb_b + 4u * (kq * RF + ci) - (SWZ ? 4u * ((kq & 1) * 16) : 0u);
                auto read_bx = [&](auto tc, float (&bx)[4]) {
                    constexpr int t = decltype(tc)::value;
                    static_for<0, 4>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (SWZ && (t & 1)) bx[q] = lds_read_raw_at<4 * (4 * q * RF + 16 * t)>(xl_o);
                        else bx[q] = lds_read_raw_at<4 * (4 * q * RF + 16 * t)>(xl_b);
                    });
                };
                if constexpr (SP == 2) {
                    // pre-split rows: element (row 4 q + kq, feature 16 t + ci) is half-word ci & 3 of the 16-byte chunk
                    // 4 t + (ci >> 2) of its row -- hi halves in the chunk's first eight bytes, lo halves in the last eight.
                    // The swizzle moves whole chunks, so the two per-lane bases of the 32-bit form carry over.
                    const uint32_t hb_b = rb_b + 4u * (kq * RF) + 16u * (ci >> 2) + 2u * (ci & 3) + (SWZ ? (kq & 1) * 64u : 0u);
                    [[maybe_unused]] const uint32_t hb_o = rb_b + 4u * (kq * RF) + 16u * (ci >> 2) + 2u * (ci & 3) - (SWZ ? (kq & 1) * 64u : 0u);
                    auto read_hx = [&](auto tc, uint32_t (&hx)[8]) {          // hx: hi of atoms q = 0..3, lo of atoms q = 0..3
                        constexpr int t = decltype(tc)::value;
                        const uint32_t base = (SWZ && (t & 1)) ? hb_o : hb_b;
                        static_for<0, 4>([&](auto qc) {
                            constexpr int q = decltype(qc)::value;
                            lds_read_h16<4 * (4 * q * RF + 16 * t)>(hx[q], base);
                            lds_read_h16<4 * (4 * q * RF + 16 * t) + 8>(hx[4 + q], base);
                        });
                    };
                    uint32_t hxp[2][8];
                    read_hx(IC<0>{}, hxp[0]);
                    static_for<0, KC>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        uint32_t (&hx)[8] = hxp[t & 1];
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hx[0]), "+v"(hx[1]), "+v"(hx[2]), "+v"(hx[3]), "+v"(hx[4]), "+v"(hx[5]),
                                     "+v"(hx[6]), "+v"(hx[7]) : : "memory");
                        if constexpr (t + 1 < KC) read_hx(IC<t + 1>{}, hxp[(t + 1) & 1]);
                        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                        SplitReg xb;                               // (whole-register writes: v_lshl_or_b32)
                        xb.hi = __builtin_bit_cast(h16x4, u32x2{hx[0] | (hx[1] << 16), hx[2] | (hx[3] << 16)});
                        xb.lo = __builtin_bit_cast(h16x4, u32x2{hx[4] | (hx[5] << 16), hx[6] | (hx[7] << 16)});
                        if constexpr (s < D) {
#pragma unroll
                            for (int b = 0; b < NBS; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(avs[b].lo, xb.hi, acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < NBS; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(avs[b].hi, xb.lo, acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < NBS; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(avs[b].hi, xb.hi, acc[b][t], 0, 0, 0);
                        } else {
                            accC[t] = split_mfma(avs[0], xb, accC[t]);
                        }
                    });
                } else if constexpr (SP) {
                    // a feature tile is three short matrix instructions per support now: the next tile's four reads are issued
                    // before this tile's conversion and products (they were hidden behind 4 x 32-cycle instructions before)
                    float bxp[2][4];
                    read_bx(IC<0>{}, bxp[0]);
                    static_for<0, KC>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        float (&bx)[4] = bxp[t & 1];
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bx[0]), "+v"(bx[1]), "+v"(bx[2]), "+v"(bx[3]) : : "memory");
                        if constexpr (t + 1 < KC) read_bx(IC<t + 1>{}, bxp[(t + 1) & 1]);
                        const SplitReg xb = split_scaled(f32x4{bx[0], bx[1], bx[2], bx[3]}, rsc);
                        if constexpr (s < D) {
#pragma unroll
                            for (int b = 0; b < NBS; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(avs[b].lo, xb.hi, acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < NBS; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(avs[b].hi, xb.lo, acc[b][t], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < NBS; ++b) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(avs[b].hi, xb.hi, acc[b][t], 0, 0, 0);
                        } else {
                            accC[t] = split_mfma(avs[0], xb, accC[t]);
                        }
                    });
                } else {
                    static_for<0, KC>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        float bx[4];
                        read_bx(tc, bx);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bx[0]), "+v"(bx[1]), "+v"(bx[2]), "+v"(bx[3]) : : "memory");
                        if constexpr (s < D) {
#pragma unroll
                            for (int b = 0; b < NBS; ++b)
#pragma unroll
                                for (int q = 0; q < 4; ++q) acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b][q], bx[q], acc[b][t], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) accC[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[q], bx[q], accC[t], 0, 0, 0);
                        }
                    });
                }
            }
            MKGNN_BPHASE(2);
            // ---- retire / barrier / issue: as in the forward
            if constexpr (RING == 2) {
                wait_vmcnt<0>();
            } else {
                static_for<0, NS>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    constexpr int n_young = young_batches<D, KC>(s, r);
                    if (role == r) wait_vmcnt<n_young>();
                });
            }
            MKGNN_BPHASE(3);
            if constexpr (NS > 1) __builtin_amdgcn_s_barrier();
            MKGNN_BPHASE(4);
            constexpr int sd = (s + RING) % S1;
            const int64_t itd = it + (s + RING) / S1;
            float* const drec = meta + (itd & 1) * META;
            issue_rows(IC<sd>{}, drec, ring + buf * SLOT);
            if constexpr (sd == 0) issue_coef(tile_at(itd), cbuf);
            if (role == 0) {
                if constexpr (sd < D) issue_bonds(tile_at(itd), sd, drec);
                if constexpr (sd == 0) {
                    issue_inv(drec);
                    issue_ids(tile_at(itd + 1), meta + ((itd + 1) & 1) * META);
                }
            }
            buf = (buf + 1 == RING) ? 0 : buf + 1;
            MKGNN_BPHASE(5);
        });
    }
    wait_vmcnt<0>();
    MKGNN_BPHASE(3);

    // ---- this wave's slice of its stream's partial slab: C layout col = feature 16 t + ci, row = kernel kq * 4 + r
    if constexpr (SP) {
        const float ug_col = split_unscale_of<0>(G);     // 1 / G of this lane's column
        float ug[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ug[r] = __shfl(ug_col, 4 * kq + r, 64);      // ... of the kernels whose rows this lane accumulates
#pragma unroll
        for (int t = 0; t < KC; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accC[t][r] *= ug[r];
#pragma unroll
                for (int b = 0; b < NBS; ++b) acc[b][t][r] *= ug[r];
            }
    }
    float* const slab = dg.slab + (size_t)sg * bank_floats(D, L, F, a.E);
    const size_t o_sup = (size_t)L * F, o_edg = o_sup + (size_t)L * D * F;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = kq * 4 + r, l = ct * kpt + i;
        if (ct_ok && i < kpt && l < L) {
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                const int f = 16 * t + ci;
                if (f < F) {
                    if (!HS || half == 0) slab[(size_t)l * F + f] = accC[t][r];
#pragma unroll
                    for (int b = 0; b < NBS; ++b) slab[o_sup + (size_t)(l * D + (HS ? 2 * half + b : b)) * F + f] = acc[b][t][r];
                }
            }
            if (ci < a.E) {
#pragma unroll
                for (int b = 0; b < NBS; ++b) slab[o_edg + (size_t)(l * D + (HS ? 2 * half + b : b)) * a.E + ci] = accE[b][r];
            }
        }
    }
#ifdef MKGNN_BWD_STAMPS
    MKGNN_BPHASE(6);
    if (g_bank_stream_stamps && lane == 0) {
        unsigned long long* o = g_bank_stream_stamps + ((size_t)blockIdx.x * 4 + wave) * 16;
        o[0] = t_start; o[1] = __builtin_readcyclecounter(); o[2] = (unsigned long long)(D * 16 + cp); o[3] = (unsigned long long)iters;
        for (int i = 0; i < 8; ++i) o[4 + i] = phase[i];
    }
#endif
}

template <int KC, int SP = 0>
__global__ void __launch_bounds__(256, (KC >= 8 ? 1 : 2)) kc_backward_bank_stream(BankStreamArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int grp = a.blk_group[blockIdx.x];
    const int rank = a.blk_rank[blockIdx.x];
    const int di = a.grp_degree[grp];
    const int cp = a.grp_cp[grp];
    const int count = a.grp_count[grp];
    switch (di) {
        case 0: bank_stream_body<1, KC, SP>(a, a.deg[0], cp, rank, count, lds); break;
        case 1: bank_stream_body<2, KC, SP>(a, a.deg[1], cp, rank, count, lds); break;
        case 2: bank_stream_body<3, KC, SP>(a, a.deg[2], cp, rank, count, lds); break;
        default: bank_stream_body<4, KC, SP>(a, a.deg[3], cp, rank, count, lds); break;
    }
}

// ---------------------------------------------------------------- host ----
// diagnostics (make STAMPS=1): device buffer of (blocks * 4 waves * 16) uint64 for the bank kernel's phase stamps; 0 = off
extern "C" int mkgnn_debug_set_bank_stream_stamps(void* device_ptr) {
#ifdef MKGNN_BWD_STAMPS
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bank_stream_stamps), &device_ptr, sizeof(void*));
#else
    (void)device_ptr;
    return -1;
#endif
}

// Same conditions as the forward's streamed kernel (the pre-pass and the slab chunk capacity added).
bool bank_stream_rows_split_supported(int F) { return (F + 15) / 16 <= 7 && bwd_split_mode() != 0; }
bool bank_stream_supported(int d, int F, int E, int L, int64_t n_atoms, int64_t x_stride, const float* e_unit) {
    return stream_forward_supported(d, F, E, L, n_atoms, x_stride, x_stride, e_unit);
}

void plan_backward_bank_stream(const BwdArgs a4[4], const bool use[4], const float* const e_unit[4], float* const coefq[4],
                               int nchunk_out[4], int ntheta_out[4], bool through_nei, BankStreamLaunch* out) {
    BankStreamArgs& a = out->a;
    memset(out, 0, sizeof(*out));
    int KC = 0, ng = 0, prep_blocks = 0;
    constexpr int MG = FUSED_MAX_GROUPS;
    double cost[MG];
    int64_t tiles_of[MG], cap[MG];
    int nstream_of[MG], deg_of[MG];
    size_t lds_fl = 0;
    for (int i = 0; i < 4; ++i) {
        nchunk_out[i] = 0; ntheta_out[i] = 0;
        if (!use[i]) continue;
        const BwdArgs& s = a4[i];
        const int d = i + 1;
        a.x = s.x; a.xs = s.xs; a.inv = s.inv; a.gout = s.gout; a.gs = s.gs; a.F = s.F; a.E = s.E;
        a.through_nei = through_nei ? 1 : 0;
        KC = (s.F + 15) / 16;
        BankStreamDeg& g = a.deg[i];
        g.sel = s.sel; g.nei = s.nei; g.e_unit = e_unit[i]; g.pair = s.pair; g.chir = s.chir; g.mix = s.mix;
        g.coefq = coefq[i]; g.slab = s.slab; g.theta_slab = s.theta_slab;
        g.n = s.n; g.L = s.L; g.off = s.off;
        g.nct = (s.L + 15) / 16;
        g.kpt = (s.L + g.nct - 1) / g.nct;
        g.cs = stream_column_parts(d, s.L);
        const int64_t ntiles = (s.n + 15) / 16;
        g.prep_blk0 = prep_blocks;
        g.prep_blocks = (int)((ntiles * g.nct + PREP_RPB - 1) / PREP_RPB);
        prep_blocks += g.prep_blocks;
        ntheta_out[i] = g.prep_blocks;
        const int nstream = d == 1 ? 4 : (d == 4 ? 1 : 2);
        const size_t fl = (size_t)bs::lds_floats(d, KC);
        if (fl > lds_fl) lds_fl = fl;
        for (int cp = 0; cp < g.cs; ++cp) {
            // a wave's time per tile (units of 32 cycles): matrix work + DMA issue, no per-tile epilogue
            const int nbs = d == 4 ? 2 : d;
            cost[ng] = (d * nbs + 1) * 4.0 * KC + 4.0 * d * nbs + 12.0 * d + 60.0;
            if (KC <= 7 && bwd_split_mode() != 0) {
                // the split-fp16 products: measured per tile (tools/bwd_stream_stamps.py, batch 4096) at F = 110 and F = 28
                static const double sp7[4] = {196.0, 317.0, 503.0, 535.0}, sp2[4] = {126.0, 193.0, 298.0, 338.0};
                cost[ng] = sp2[i] + (sp7[i] - sp2[i]) * (KC - 2) / 5.0;
            }
            tiles_of[ng] = ntiles;
            cap[ng] = (ntiles + nstream - 1) / nstream;
            nstream_of[ng] = nstream;
            deg_of[ng] = i;
            a.grp_degree[ng] = (uint8_t)i;
            a.grp_cp[ng] = (uint8_t)cp;
            ++ng;
        }
    }
    if (ng == 0) return;
    // block counts: greedy min-max as in the forward; the column parts of a degree get the same count (they fill the
    // same slab chunks), and a degree's streams may not outnumber its slab chunks
    auto finish = [&](int g, int blocks) {
        const int64_t streams = (int64_t)blocks * nstream_of[g];
        return 40.0 + (double)((tiles_of[g] + streams - 1) / streams) * cost[g];
    };
    int count[MG], nb = 0;
    for (int g = 0; g < ng; ++g) { count[g] = 1; ++nb; }
    auto parts_of = [&](int g) {                         // the groups of g's degree (its column parts) grow together
        int n = 0;
        for (int h = 0; h < ng; ++h) if (deg_of[h] == deg_of[g]) ++n;
        return n;
    };
    // Grid cap.  Not every wave slot of the chip: this kernel runs beside the other chain's kernels (the gather, the next
    // layer's rows kernel), which need slots to make progress at all -- 320 blocks instead of 512 is 2.5 % of the step at
    // batch 4096 (measured together with the rows kernel's 448; MKGNN_BANK_STREAM_BLOCKS / MKGNN_ROWS_STREAM_BLOCKS to re-measure)
    static const char* env_blocks = getenv("MKGNN_BANK_STREAM_BLOCKS");
    const int max_blocks = grid_cap(g_grid_caps.bank, env_blocks && atoi(env_blocks) > 8 && atoi(env_blocks) <= FUSED_MAX_BLOCKS ? atoi(env_blocks) : 320);
    while (nb < max_blocks) {
        int worst = -1;
        double t_worst = -1.0;
        for (int g = 0; g < ng; ++g) {
            if (count[g] >= cap[g] || (int64_t)(count[g] + 1) * nstream_of[g] > SLAB_CHUNKS) continue;
            const double t = finish(g, count[g]);
            if (t > t_worst) { t_worst = t; worst = g; }
        }
        if (worst < 0) break;
        const int np = parts_of(worst);
        if (nb + np > max_blocks) break;
        for (int h = 0; h < ng; ++h) if (deg_of[h] == deg_of[worst]) ++count[h];
        nb += np;
    }
    int given[MG] = {};
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
        int per_xcd[MG][8] = {};
        for (int b = 0; b < nb; ++b) ++per_xcd[a.blk_group[b]][b & 7];
        int next[MG][8];
        for (int g = 0; g < ng; ++g) {
            int run = 0;
            for (int x = 0; x < 8; ++x) { next[g][x] = run; run += per_xcd[g][x]; }
        }
        for (int b = 0; b < nb; ++b) a.blk_rank[b] = (uint16_t)next[a.blk_group[b]][b & 7]++;
    }
    for (int g = 0; g < ng; ++g) {
        a.grp_count[g] = (uint16_t)count[g];
        nchunk_out[deg_of[g]] = count[g] * nstream_of[g];
    }
    note_plan(2, nb, ng, tiles_of, count, nstream_of);
    out->nb = nb; out->prep_blocks = prep_blocks; out->KC = KC; out->lds_bytes = lds_fl * 4;
}

// the pre-pass: its records feed the rows kernel and the bank kernel
hipError_t launch_coef_prepare(const BankStreamLaunch& p, hipStream_t st) {
    if (p.prep_blocks == 0) return hipSuccess;
    coef_prepare_kernel<<<p.prep_blocks, 256, 0, st>>>(p.a);
    return hipGetLastError();
}

template <int KC, int SP = 0> static hipError_t launch_bank_kc(const BankStreamLaunch& p, hipStream_t st) {
    if (p.lds_bytes > 64 * 1024) {
        static PerDeviceOnce attr_set;
        if (const int slot = attr_set.pending(); slot >= 0) {
            hipError_t e = hipFuncSetAttribute((const void*)kc_backward_bank_stream<KC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (KC >= 8 ? 128 : 80) * 1024);
            if (e != hipSuccess) return e;
            attr_set.set(slot);
        }
    }
    kc_backward_bank_stream<KC, SP><<<p.nb, 256, p.lds_bytes, st>>>(p.a);
    return hipGetLastError();
}

hipError_t launch_backward_bank_stream(const BankStreamLaunch& p, hipStream_t st) {
    if (p.nb == 0) return hipSuccess;
    if (p.lds_bytes > (size_t)(p.KC >= 8 ? 128 : 80) * 1024) return hipErrorInvalidValue;
    g_last_plan[2].launches.fetch_add(1);
    if (p.x_split) {                                     // pre-split atom rows (the caller has asked bank_stream_rows_split_supported)
        if (!(p.KC <= 7 && bwd_split_mode() != 0)) return hipErrorInvalidValue;
        switch (p.KC) {
            case 1: return launch_bank_kc<1, 2>(p, st);
            case 2: return launch_bank_kc<2, 2>(p, st);
            case 3: return launch_bank_kc<3, 2>(p, st);
            case 4: return launch_bank_kc<4, 2>(p, st);
            case 5: return launch_bank_kc<5, 2>(p, st);
            case 6: return launch_bank_kc<6, 2>(p, st);
            default: return launch_bank_kc<7, 2>(p, st);
        }
    }
    if (p.KC <= 7 && bwd_split_mode() != 0) {
        switch (p.KC) {
            case 1: return launch_bank_kc<1, true>(p, st);
            case 2: return launch_bank_kc<2, true>(p, st);
            case 3: return launch_bank_kc<3, true>(p, st);
            case 4: return launch_bank_kc<4, true>(p, st);
            case 5: return launch_bank_kc<5, true>(p, st);
            case 6: return launch_bank_kc<6, true>(p, st);
            default: return launch_bank_kc<7, true>(p, st);
        }
    }
    switch (p.KC) {
        case 1: return launch_bank_kc<1>(p, st);
        case 2: return launch_bank_kc<2>(p, st);
        case 3: return launch_bank_kc<3>(p, st);
        case 4: return launch_bank_kc<4>(p, st);
        case 5: return launch_bank_kc<5>(p, st);
        case 6: return launch_bank_kc<6>(p, st);
        case 7: return launch_bank_kc<7>(p, st);
        case 8: return launch_bank_kc<8>(p, st);
        case 9: return launch_bank_kc<9>(p, st);
        case 10: return launch_bank_kc<10>(p, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mkgnn
