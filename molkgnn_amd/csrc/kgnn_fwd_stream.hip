// Streamed MFMA forward of the kernel convolution (gfx950): ONE launch = one KernelSetConv forward
// (reference kernels.py:610-751 with :353-425 fused), like kc_forward_fused, restructured so that the inner loop
// contains nothing but matrix instructions and one LDS read per 4 D of them.
//
// What kc_forward_fused (kgnn_mfma.hip) spends its time on besides the matrix pipe: D LDS reads of the bank per chunk
// (each chunk step its own basic block: the compiler cannot hoist them over the step's branch), gathered rows held in
// 28 - 56 VGPRs per wave, every degree-3 / 4 tile gathered once per column part.  Here the two operands swap places:
//
//   * the BANK is the register-resident operand.  A wave owns one column tile (<= 16 kernels) of its degree and keeps
//     those kernels' unit rows -- D support slots + the centre, KC 16-byte chunks each -- in (D + 1) * KC * 4 VGPRs for
//     the whole launch (140 for degree 4).  Nothing about the bank is read inside the loop.
//   * the gathered ATOM ROWS stream through LDS.  The NS waves that hold the NS column tiles of a degree share one
//     16-atom tile: its rows are fetched ONCE, by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, asynchronous),
//     slot by slot (neighbour 0 .. D-1, then the focal row) into a ring of KC KB buffers laid out in exactly the order
//     the MFMA A-operand lanes read them (lane q of DMA piece t fetches chunk 4 t + (q >> 4) of row q & 15 and lands at
//     byte 16 q of piece t: the read is the linear ds_read_b128 at 16 * lane, conflict-free by construction).
//     One ds_read_b128 feeds 4 * D MFMAs; the ring runs RING slots ahead of the multiply.
//   * per-tile small data (atom ids, 1 / |x| of every slot, focal ids, raw bond attributes, the degree-4 chirality
//     flags) is DMA'd / written into a small LDS record in the layout its readers use, a tile ahead.
//
// Synchronisation (cdna_hip_programming.md section 5, "Pipelining across barriers"): every vector-memory LOAD of the
// loop is an LDS-DMA with a hand-counted s_waitcnt vmcnt(N) -- the compiler never sees an ordinary load result in the
// loop, so it never drains the DMA queue with a vmcnt(0) of its own -- and waves meet at a raw s_barrier.  Step k (one
// slot of one tile):  multiply buffer k % RING;  wait until only the youngest DMA batch is in flight (so the batch for
// step k + 1 has landed);  barrier (all waves are done reading buffer k % RING, everybody's share of step k + 1 is
// visible);  issue the DMA batch of step k + RING into buffer k % RING;  after a tile's last slot, the epilogue
// (permutation maximum, bond cosines, mixing, chirality sign, stores) -- the arithmetic of kc_forward_fused, so the two
// kernels agree bit for bit.
//
// Degree 4 (140 bank registers + 64 accumulators per wave would spill, and a scratch reload is a vector-memory load that
// drains the DMA queue): a column tile's four support slots are split over TWO waves (supports {0,1} + the centre, and
// {2,3}); a block holds two column tiles, the degree's four take two blocks (its rows are gathered twice -- 19 MB more
// of ~140 MB).  At the end of a tile the two waves swap, through LDS, the half of the 4 x 4 cosine matrices the other
// needs: each finishes two of a lane's four atoms, with the same summation order as everywhere else.
//
// Covered shapes (round 3): any F <= 160 (KC = ceil(F / 16) = 1 .. 10 sixteen-float chunks per row; the reference's 28 and
// 110 are KC = 2 and 7, its (5, 10, 15, 25) banks give F = 55 = KC 4, a (1, 1, 1, 1) sweep F = 4 = KC 1, (16, 32, 48, 64) banks
// F = 160 = KC 10; from KC = 8 on the bank needs more than half the register file: one wave per SIMD), E <= 8, 16-byte
// aligned rows, and ANY number of kernels per degree: a stream of NS(d) = 1 / 2 / 2 / 4 waves holds CT(d) = 1 / 2 / 2 / 2
// column tiles (<= 16 kernels each), a degree with more column tiles than that is cut into column PARTS, each its own
// group of blocks that gathers the atom rows again (what degree 4's 50 kernels always did); a part with fewer column
// tiles than CT(d) leaves a wave with nothing but zeros to multiply.  The reference's 10 / 20 / 30 / 50 fill every wave.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kgnn_launch.h"
#include "kgnn_split.h"

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D> struct StreamTraits {
    static constexpr int NS = (D == 1) ? 1 : (D == 4 ? 4 : 2);     // waves sharing an atom tile = column tiles of the degree
    static constexpr int NSTREAM = 4 / NS;                         // atom-tile streams per 4-wave block
    // row-slot buffers per stream (<= S1): the batch of step k + RING is issued when step k has been multiplied, so the
    // rows of a slot have RING - 1 steps of matrix work (0.9 - 1.7 us each at two waves per SIMD) to arrive
    static constexpr int RING = (D == 1) ? 2 : (D == 2 ? 3 : (D == 3 ? 4 : 5));
    static constexpr int S1 = D + 1;                               // steps (slots) per tile
    // per-tile record (floats): ids[S1][16] | inv[S1][16] | focal[16] | sign bytes[16] (4 dwords used) | eq bytes[16] | bond[D][16][8]
    static constexpr int OFF_INV = 16 * S1, OFF_FOCAL = 32 * S1, OFF_SIGN = OFF_FOCAL + 16, OFF_EQ = OFF_SIGN + 16,
                         OFF_BOND = OFF_EQ + 16, META = OFF_BOND + 128 * D;
};

int stream_tiles_per_part(int d) { return d == 1 ? 1 : 2; }          // CT(d): column tiles a stream's waves hold
int stream_column_parts(int d, int L) { return ((L + 15) / 16 + stream_tiles_per_part(d) - 1) / stream_tiles_per_part(d); }

__host__ __device__ constexpr int stream_lds_floats(int D, int KC) {
    const int NS = (D == 1) ? 1 : (D == 4 ? 4 : 2);
    const int RING = (D == 1) ? 2 : (D == 2 ? 3 : (D == 3 ? 4 : 5));
    const int META = 32 * (D + 1) + 48 + 128 * D;
    // degree 4: + chirality sign table (<= 64 x 12 bytes) + the exchange buffer of the support halves (4 waves x 18 KB-rows of 256 B)
    return (4 / NS) * (RING * KC * 256 + 2 * META) + (D == 4 ? 192 + 4 * 18 * 64 : 0);
}

// Ping-pong launch (PP, round 5): a block is TWO such 4-wave teams ("sides"), each with its own rings and records; the
// chirality table is shared, the exchange buffer holds all 8 waves.
__host__ __device__ constexpr int stream_side_floats(int D, int KC) {
    const int NS = (D == 1) ? 1 : (D == 4 ? 4 : 2);
    const int RING = D + 1;
    const int META = 32 * (D + 1) + 48 + 128 * D;
    return (4 / NS) * (RING * KC * 256 + 2 * META);
}
__host__ __device__ constexpr int stream_pp_lds_floats(int D, int KC) {
    return 2 * stream_side_floats(D, KC) + (D == 4 ? 192 + 8 * 18 * 64 : 0);
}

template <int I> using IC = std::integral_constant<int, I>;
template <int B, int E, typename Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (B < E) { fn(IC<B>{}); static_for<B + 1, E>(fn); }
}

// pi_p(a) of the reference's permutation tables (kernels.py:109-128) without a table load: the a-th entries of all
// orders packed two bits each (a is a compile-time constant wherever this is used; a __constant__ lookup with a
// per-lane index is an ordinary vector load, which the DMA pipeline of this kernel cannot afford)
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

// Diagnostic build (make STAMPS=1): per-wave cycle totals of the loop's phases -- multiply, counted wait, barrier, DMA issue,
// epilogue -- in stamps[4..8] (tools/stream_stamps.py).  Each s_memtime read costs an lgkmcnt(0) at a phase boundary.
#ifdef MKGNN_FWD_STAMPS
#define MKGNN_PHASE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); phase[i] += t_ - t_phase; t_phase = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MKGNN_PHASE(i) do { } while (0)
#endif

// Wave priority of the multiply phase / of everything else (waits, DMA issue, epilogue).  Two waves share a SIMD: the one
// that is multiplying needs one issue slot per 32 cycles, the other one's VALU-and-latency-bound epilogue decides how soon
// it multiplies again (MI355X_MICROARCH.md, "Two waves per SIMD", items 2 and 4).  -DMKGNN_STREAM_PRIO=<0|1|2>: A/B builds.
#ifndef MKGNN_STREAM_PRIO
#define MKGNN_STREAM_PRIO 1
#endif
__device__ __forceinline__ void prio_multiply() {
    if constexpr (MKGNN_STREAM_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    else if constexpr (MKGNN_STREAM_PRIO == 2) __builtin_amdgcn_s_setprio(2);
}
__device__ __forceinline__ void prio_other() {
    if constexpr (MKGNN_STREAM_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    else if constexpr (MKGNN_STREAM_PRIO == 2) __builtin_amdgcn_s_setprio(0);
}

// LDS reads behind the compiler's back (immediate byte offset, one address register per family of reads) and the wait that
// hands their registers over
template <int OFFB> __device__ __forceinline__ void lds_read128(f32x4& dst, uint32_t addr) {
    static_assert(OFFB >= 0 && OFFB < 65536 && OFFB % 16 == 0, "ds_read_b128 offset: 16 bits, 16-byte aligned");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFFB));
}
template <int OFFB> __device__ __forceinline__ void lds_read32(float& dst, uint32_t addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFFB) : "memory");
}
__device__ __forceinline__ void lds_wait0(float& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void lds_after_wait(float& a) { asm volatile("" : "+v"(a) : : "memory"); }      // (a value read before an lds_wait0 of another)
__device__ __forceinline__ void lds_wait0(f32x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void lds_wait0(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "s_waitcnt vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16 / 4 bytes per lane, global -> LDS, asynchronous; the LDS address is the wave-uniform base + lane * size.
//
// SITE: every place that issues an LDS-DMA gets its own instruction offset (subtracted from both pointers first: the
// hardware adds it to the memory address AND to the LDS address -- tools/probes/dma_offset_probe.hip -- so nothing
// moves).  It is there for the compiler: two calls of the intrinsic that differ only in their (run-time) pointers are
// "the same instruction" to LLVM's code sinking, which then merges a DMA under `if (lane < 32)` with the unconditional one
// that follows it into ONE instruction whose LDS base is a phi of the two destinations -- made wave-uniform by
// v_readfirstlane, i.e. half of the lanes' rows landed in the other site's record (KC = 1, where a slot is a single piece
// next to the bond rows' DMA: every neighbour slot's upper k-lanes were wrong).  Immediate operands cannot be phi'd.
#define MKGNN_DMA_CASE(SZ, O) else if constexpr (OFF == O) __builtin_amdgcn_global_load_lds(g, l, SZ, O, 0)
template <int OFF> __device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
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
    MKGNN_DMA_CASE(4, 4); MKGNN_DMA_CASE(4, 8); MKGNN_DMA_CASE(4, 12); MKGNN_DMA_CASE(4, 16); MKGNN_DMA_CASE(4, 20);
    else static_assert(OFF < 0, "add the offset to the list");
}
constexpr int SITE_ROWS = 0, SITE_BONDS = 16;                                     // dma16
constexpr int SITE_IDS = 0, SITE_IDS4 = 4, SITE_INV = 8, SITE_INV4 = 12, SITE_SIGN = 16, SITE_EQ = 20;      // dma4

// DMA pieces of one row slot that wave `role` (0 .. NS-1) of a stream issues: t = role, role + NS, ...
template <int KC, int NS> constexpr int pieces_of(int role) { return (KC - role + NS - 1) / NS; }

// vector-memory operations wave `role` issues in the DMA phase whose slot is `sd` (the counted waits rely on it)
template <int D, int KC> constexpr int batch_size(int sd, int role) {
    using T = StreamTraits<D>;
    int n = pieces_of<KC, T::NS>(role);
    if (role == 0) {
        if (sd < D) n += 1;                              // unit bond rows of the slot
        if (sd == 0) n += 2 * (T::S1 == 5 ? 2 : 1) + (D == 4 ? 2 : 0);     // following tile's ids, this tile's 1/|x|, flags
    }
    return n;
}

// vector-memory operations of the RING - 2 batches issued at the ends of steps k-1 .. k-(RING-2), k being a step with slot s
template <int D, int KC> constexpr int young_batches(int s, int role) {
    using T = StreamTraits<D>;
    int n = 0;
    for (int j = 1; j <= T::RING - 2; ++j) n += batch_size<D, KC>((s + 64 * T::S1 - j + T::RING) % T::S1, role);
    return n;
}

// BF = true: the bf16 similarity path (BASELINE configs[4]) on the streamed skeleton -- round 4.  The node-feature dot products
// take bf16 operands and accumulate in fp32: a lane's four consecutive columns of a 16-float chunk are exactly the k = 4 (lane >> 4)
// + i of v_mfma_f32_16x16x16_bf16, so the atom chunk read from the ring is converted (two v_cvt_pk_bf16_f32) and ONE matrix
// instruction of 8 cycles replaces four of 32; the bank sits in the registers as bf16 (half of them).  Norms, bond cosines,
// the order scan, mix and stores are the fp32 code.  (Rounds 1-3 ran this variant on the LDS-bank kernel of kgnn_mfma.hip.)
typedef short s16x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4s to_bf16x4s(f32x4 v) {
    // (plain conversions: the compiler emits v_cvt_pk_bf16_f32 and knows the VALU -> MFMA operand hazards)
    const bf16x4s r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(s16x4s, r);
}

// BF = 2 (round 5): fp32 products out of fp16 matrix instructions -- the exact hi + lo split of kgnn_split.h; bank rows are unit
// rows, scaled by 2^8 before the split; an atom's row (a lane's A operand belongs to ONE atom: lane & 15) is scaled by
// 2^(exponent(1 / |x|) + 8), i.e. to a norm in [256, 512), and the two powers of two leave through the 1 / |x| factor the
// epilogue multiplies with anyway (exponent field set to -16: exact).
constexpr int SPLIT_BANK_EXP = 8, SPLIT_ROW_EXP = 8;
// 2^(exponent(inv) + SPLIT_ROW_EXP): the scale of an atom's row (inv = 1 / max(|x|, eps) <= 1e8: no overflow)
__device__ __forceinline__ float split_row_scale(float inv) {
    return __uint_as_float((__float_as_uint(inv) & 0x7f800000u) + ((uint32_t)SPLIT_ROW_EXP << 23));
}
// inv / (row scale * bank scale): inv's mantissa with the exponent -(SPLIT_ROW_EXP + SPLIT_BANK_EXP)
__device__ __forceinline__ float split_inv(float inv) {
    return __uint_as_float((__float_as_uint(inv) & 0x007fffffu) | ((uint32_t)(127 - SPLIT_ROW_EXP - SPLIT_BANK_EXP) << 23));
}

// PP = true (round 5, "ping-pong"): 8-wave blocks, one per CU.  Waves 0-3 ("side 0") and 4-7 ("side 1") are two teams of the
// 4-wave kind above, each on its own tiles; wave w and wave w + 4 share a SIMD.  Measured with ONE wave per SIMD
// (tools/stream_stamps.py --blocks 256) a tile costs a degree-3 wave 10.4 k cycles of products and 7.2 k of everything else
// (DMA issue, order scan, bond cosines, mix, stores); two unsynchronised waves per SIMD take 26.5 k each -- 13.2 k per tile
// and SIMD, the matrix pipe 68 % busy in the steady state -- because their product phases collide as often as they
// interleave.  Here they cannot collide: a side's tile is two PHASES separated by block-wide barriers,
//     M: multiply all S1 slots of the tile (the ring holds exactly one tile: RING = S1) -- matrix instructions + LDS reads only;
//     O: issue the whole next tile's DMA into the ring just read, then the epilogue of this tile, then vmcnt(0);
// and side 1 runs one phase behind side 0, so on every SIMD one wave multiplies while its partner issues, scans and stores.
// No counted vmcnt any more (every DMA of a tile lands before the phase that reads it begins).  Degree 4 (two waves per
// column tile swap halves through LDS in the epilogue, with a barrier): every phase has a second barrier in its middle.
template <int D, int KC, int BF = 0, bool PP = false>
__device__ __forceinline__ void stream_body(const FusedFwdArgs& a, const FusedDeg& dg, const int cp, const int rank, const int count, float* lds) {
    using T = StreamTraits<D>;
    constexpr int NS = T::NS, NSTREAM = T::NSTREAM, RING = T::RING, S1 = T::S1, META = T::META;
    static_assert(!PP || RING == S1, "ping-pong: the ring holds exactly one tile");
    constexpr int NSIDE = PP ? 2 : 1;
    constexpr int NTHREADS = PP ? 512 : 256;
    constexpr int SLOT = KC * 256;                       // floats per row-slot buffer (KC pieces of 1 KB)
    // HS ("half supports", degree 4): wave = (column tile of the block, support half); see the file comment
    constexpr bool HS = (D == 4);
    constexpr int NBS = HS ? 2 : D;                      // support slots this wave multiplies
    constexpr int NB = NBS + 1;                          // bank slots in registers: the supports + the centre (HS: half 0 only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave of the block (PP: 0..7)
    const int side = PP ? (wave8 >> 2) : 0;                           // PP: team of the block
    const int wave = PP ? (wave8 & 3) : wave8;                        // wave of its team
    const unsigned long long t_start = a.stamps ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long rt_start = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz: the in-kernel clock is d(cycles) / d(this)
    const int stream = wave / NS, role = wave % NS;
    const int half = HS ? (role & 1) : 0;                // supports 2 * half, 2 * half + 1
    const int ct = HS ? cp * 2 + (role >> 1) : cp * NS + role;     // this wave's column tile (past the degree's last: an idle wave)
    const int ci = lane & 15, kq = lane >> 4;
    const int L = dg.L, kpt = dg.kpt;
    const int lcol = ct * kpt + ci;
    const bool col_ok = (ct < dg.nct) && (ci < kpt) && (lcol < L);
    const int FPB = a.FPB;                               // row pitch of the padded bank (>= FP)
    const bool do_chir = (D == 4) && a.last;
    constexpr int SIDE_FLOATS = NSTREAM * (RING * SLOT + 2 * META);
    float* const ring = lds + (size_t)side * SIDE_FLOATS + (size_t)stream * (RING * SLOT + 2 * META);
    float* const meta = ring + RING * SLOT;
    int8_t* const chirtab = (int8_t*)(lds + (size_t)NSIDE * SIDE_FLOATS);
    float* const xbuf = lds + (size_t)NSIDE * SIDE_FLOATS + 192;      // (HS only) [wave of the block][18][64]

    // ---- tiles of this stream: a contiguous run; every block of the group runs the same number of iterations
    // (32-bit from here on: the host admits n_atoms * stride < 2^30 only, and wave-uniform 64-bit values are what this kernel's
    // scalar registers spill on)
    const int dn = (int)dg.n;
    const int ntiles = (dn + 15) / 16;
    const int nstreams = count * NSTREAM * NSIDE;
    const int sg = (rank * NSIDE + side) * NSTREAM + stream;
    auto run_start = [&](int g) -> int { return (int)((uint64_t)(uint32_t)g * (uint32_t)ntiles / (uint32_t)nstreams); };    // < 2^11 streams x < 2^26 tiles
    const int tile_first = run_start(sg);
    const int tile_end = run_start(sg + 1);
    // iterations of this BLOCK: the most tiles any of its streams owns (its waves meet at the same barriers; a stream with one
    // tile fewer repeats its last one, results discarded).  Until round 5 every block of a group ran the group's maximum:
    // at batch 4096 the degree-2 group ran 3 150 tile slots for 2 875 tiles.
    int iters = 0;
    {
        const int sg0 = rank * NSIDE * NSTREAM;
#pragma unroll
        for (int q = 0; q < NSIDE * NSTREAM; ++q) {
            const int c = run_start(sg0 + q + 1) - run_start(sg0 + q);
            iters = c > iters ? c : iters;
        }
        if (iters < 1) iters = 1;
    }
    const int tile_hi = (tile_end > tile_first ? tile_end : (tile_first + 1 < ntiles ? tile_first + 1 : ntiles)) - 1;
    auto tile_at = [&](int i) -> int {                   // clamped: a stream short of tiles repeats its last one
        const int t = tile_first + i;
        return t > tile_hi ? tile_hi : t;
    };

    // PP: the team on the other side of the block (same stream, same role: its waves are this team's SIMD partners)
    [[maybe_unused]] float* const pring = lds + (size_t)(PP ? 1 - side : 0) * SIDE_FLOATS + (size_t)stream * (RING * SLOT + 2 * META);
    [[maybe_unused]] float* const pmeta = pring + RING * SLOT;
    const int psg = (rank * NSIDE + (PP ? 1 - side : 0)) * NSTREAM + stream;
    const int ptile_first = run_start(psg), ptile_end = run_start(psg + 1);
    const int ptile_hi = (ptile_end > ptile_first ? ptile_end : (ptile_first + 1 < ntiles ? ptile_first + 1 : ntiles)) - 1;
    [[maybe_unused]] auto ptile_at = [&](int i) -> int {
        const int t = ptile_first + i;
        return t > ptile_hi ? ptile_hi : t;
    };

    // (tile 0's ids first: their DMA flies while the bank is loaded -- one dependent round trip less in front of the first
    // tile, which is all a stream has at small batches)
    if (role == 0) {
        int n0 = tile_at(0) * 16 + (lane & 15);
        if (n0 >= dn) n0 = dn - 1;
        const int kq0 = lane >> 4;
        const void* src0 = (kq0 < D) ? (const void*)(dg.nei + ((uint32_t)n0 * D + kq0)) : (const void*)(dg.sel + n0);
        float* const meta0 = meta;
        if (S1 >= 4 || kq0 < S1) dma4<SITE_IDS>(src0, meta0);
        if constexpr (S1 == 5) {
            if (kq0 == 0) dma4<SITE_IDS4>(dg.sel + n0, meta0 + 64);
        }
    }
    // ---- one-time: this wave's share of the bank -> registers (unit rows, zero beyond F; idle columns all zero)
    static_assert(BF == 0 || BF == 1 || ((BF == 2 || BF == 3) && !PP),
                  "BF: 0 fp32, 1 bf16 similarity, 2 fp32 out of split fp16 (4-wave blocks), 3 the same on pre-split atom rows");
    constexpr bool SPL = (BF == 2 || BF == 3);           // split-fp16 products
    constexpr bool PRE = (BF == 3);                      // the atom rows arrive pre-split (kgnn_split.h, split_row_store): no conversion here
    using BankReg = std::conditional_t<SPL, SplitReg, std::conditional_t<BF == 1, s16x4s, f32x4>>;
    BankReg bk[NB][KC];
    float2 bv[D];
    {
        const int l = col_ok ? lcol : 0;
#pragma unroll
        for (int bl = 0; bl < NB; ++bl) {
            // bank slot behind register slot bl: support 2 * half + bl (HS) or bl, the centre (slot D) last
            const int b = (bl == NBS) ? D : (HS ? 2 * half + bl : bl);
            const bool zero = !col_ok || (HS && bl == NBS && half != 0);      // (the second half multiplies no centre)
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                f32x4 v = *(const f32x4*)(dg.padded + ((size_t)b * L + l) * FPB + 16 * t + 4 * kq);
                if (zero) v = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (SPL) bk[bl][t] = split_scaled(v, (float)(1 << SPLIT_BANK_EXP));     // (10 instructions; the compiler's split_f16: 17)
                else if constexpr (BF == 1) bk[bl][t] = to_bf16x4s(v);
                else bk[bl][t] = v;
            }
        }
#pragma unroll
        for (int b = 0; b < D; ++b) {
            float2 v = *(const float2*)(dg.edge_padded + ((size_t)b * L + l) * 8 + 2 * kq);
            if (!col_ok) v = float2{0.f, 0.f};
            bv[b] = v;
        }
    }
    const float ws = dg.mix[0], wc = dg.mix[1], we = dg.mix[2], wsum = dg.mix[3];
    if constexpr (D == 4) {
        if (do_chir)
            for (int q = tid; q < L * 12; q += NTHREADS) chirtab[q] = dg.chir[q];
    }
    const int8_t* const eqp = do_chir ? (const int8_t*)dg.eqflag : (const int8_t*)dg.sel;       // always loadable
    const int8_t* const sgp = do_chir ? (const int8_t*)dg.signflag : (const int8_t*)dg.sel;
    const uint32_t xs = (uint32_t)a.xs;
    const int F = a.F;

    uint32_t idsD[S1];                                   // atom ids (of atom ci, every slot) of the tile the DMA pointer is in
    auto issue_rows = [&](auto sdc, float* buf) {        // this wave's pieces of slot sd of the DMA tile
        constexpr int sd = decltype(sdc)::value;
        const uint32_t rowbase = idsD[sd] * xs + 4u * kq;     // 32-bit element offset (the host checks n_atoms * stride < 2^30)
        static_for<0, NS>([&](auto rc) {                 // one wave-uniform branch per step, the role's pieces inside
            constexpr int r = decltype(rc)::value;
            if (role == r) {
                static_for<0, KC>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if constexpr (t % NS == r) {
                        // a chunk entirely beyond the row's width is fetched from the row's first chunk instead (in
                        // bounds, finite whenever the row is) and masked to zero where it is used
                        const uint32_t off = (t == KC - 1 && 16 * t + 4 * kq >= F) ? idsD[sd] * xs : rowbase + 16u * t;
                        dma16<SITE_ROWS>(a.x + off, buf + t * 256);
                    }
                });
            }
        });
    };
    // unit bond rows of slot sd (mkgnn_degree_bucket.nei_edge_unit, [N_d, d, 8]): 16 atoms x 32 bytes = half a DMA piece
    auto issue_bonds = [&](int t, int sd, float* mrec) {
        if (lane < 32) {
            int n = t * 16 + (lane >> 1);
            if (n >= dn) n = dn - 1;
            dma16<SITE_BONDS>(dg.e_unit + (((uint32_t)n * D + sd) * 8u + 4u * (lane & 1)), mrec + T::OFF_BOND + sd * 128);
        }
    };
    // lane q -> (slot q >> 4, atom q & 15): slots 0..3 in one 64-dword piece, slot 4 (degree 4's focal row) in a second
    auto issue_ids = [&](int t, float* mrec) {           // low dwords of the int64 indices of tile t
        int n = t * 16 + ci;
        if (n >= dn) n = dn - 1;
        const void* src = (kq < D) ? (const void*)(dg.nei + ((uint32_t)n * D + kq)) : (const void*)(dg.sel + n);    // kq == D: the focal id (D < 4)
        if (S1 >= 4 || kq < S1) dma4<SITE_IDS>(src, mrec);
        if constexpr (S1 == 5) {
            if (kq == 0) dma4<SITE_IDS4>(dg.sel + n, mrec + 64);
        }
    };
    auto issue_meta = [&](int t, float* mrec) {          // 1 / |x| of every slot (through the ids in the record), flags, focal ids
        // lane q needs the id of (slot q >> 4, atom q & 15): read it from the record (idsD[kq] would be a run-time index)
        const uint32_t idq = __float_as_uint(mrec[16 * ((S1 >= 4 || kq < S1) ? kq : 0) + ci]);
        if (S1 >= 4 || kq < S1) dma4<SITE_INV>(a.inv + idq, mrec + T::OFF_INV);
        if constexpr (S1 == 5) {
            if (kq == 0) dma4<SITE_INV4>(a.inv + idsD[4], mrec + T::OFF_INV + 64);
        }
        if constexpr (D == 4) {                          // 16 flag bytes each = 4 dwords
            if (lane < 4) {
                int n4 = t * 4 + lane;                   // dword index; the last tile may be partial: clamp into the array
                const int hi = (dn + 3) / 4 - 1;
                n4 = n4 > hi ? hi : n4;
                dma4<SITE_SIGN>(sgp + 4 * n4, mrec + T::OFF_SIGN);
                dma4<SITE_EQ>(eqp + 4 * n4, mrec + T::OFF_EQ);
            }
        }
        if (kq == 0) mrec[T::OFF_FOCAL + ci] = __uint_as_float(idsD[D]);     // (a plain LDS store from registers)
    };

    // ---- prologue: tile 0's ids go through the LDS record like every later tile's (idsD must never be indexed by a
    // run-time value: the array would live in scratch); then the first RING slots, tile 0's record, tile 1's ids;
    // everything drained before the loop
    wait_vmcnt<0>();                                     // (tile 0's ids: issued ahead of the bank load)
    __syncthreads();                                     // (also: chirality table written)
#pragma unroll
    for (int q = 0; q < S1; ++q) idsD[q] = __float_as_uint(meta[16 * q + ci]);
    static_for<0, RING>([&](auto kc) {                   // RING <= S1: all inside tile 0
        constexpr int k = decltype(kc)::value;
        issue_rows(kc, ring + k * SLOT);
        if (role == 0 && k < D) issue_bonds(tile_at(0), k, meta);
    });
    if (role == 0) {
        issue_meta(tile_at(0), meta);
        issue_ids(tile_at(1), meta + META);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    [[maybe_unused]] int buf = 0;                        // ring buffer of the current step
    if constexpr (PP) {                                  // side 1 runs one phase behind: it multiplies while side 0 finishes a tile
        if (side == 1) {
            __builtin_amdgcn_s_barrier();
            if constexpr (HS) __builtin_amdgcn_s_barrier();
        }
    }
#ifdef MKGNN_FWD_STAMPS
    unsigned long long phase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_phase = __builtin_readcyclecounter();
    phase[7] = t_phase - t_start;                        // the prologue: bank into registers, tile 0's ids, rows and record
#endif
    for (int it = 0; it < iters; ++it) {
        const int tile = tile_at(it);
        const bool real = tile_first + it < tile_end;    // else: a repeat of the last tile, results discarded
        float* const mrec = meta + (it & 1) * META;
        f32x4 cm[D][NBS];
        f32x4 cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < D; ++s)
#pragma unroll
            for (int b = 0; b < NBS; ++b) cm[s][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PP) {
            // ---- M phase: the whole tile.  Slots 0 .. D-2 one after the other; the last neighbour slot and the focal slot
            // chunk by chunk TOGETHER, so that two consecutive matrix instructions never share an accumulator (the centre's
            // chain alone would issue every 40 cycles instead of every 32: dependent-accumulator latency; every sum keeps
            // its own order).
            // The DMA of the NEXT tile is issued from here too -- but the PARTNER team's, not this team's: this team's ring
            // is being read, the partner team (one phase behind / ahead) has just finished reading its own.  A first version
            // issued a team's whole next tile at the head of its own O phase: 17-20 pieces in one burst from every wave of
            // the phase cost ~500 cycles of issue EACH (7 k cycles per tile, tools/stream_stamps.py) -- the request queue is
            // a few pieces deep and an issue blocks while it is full.  One or two pieces between two chunks of matrix
            // instructions (8-12 of them, 256-384 cycles) find the queue drained.
            const int pt = it + side;                    // the partner's tile (iteration) whose rows are fetched in this phase
            {
                const float* const prec = pmeta + (pt & 1) * META;
#pragma unroll
                for (int q = 0; q < S1; ++q) idsD[q] = __float_as_uint(prec[16 * q + ci]);      // (its ids were DMA'd a tile ago)
            }
#ifndef MKGNN_PP_NO_DMA                                   // (timing experiment: nothing is fetched after tile 0)
            if (role == 0) {                             // the small records first (bond rows, 1 / |x| and flags, the ids of the tile after)
                float* const prec = pmeta + (pt & 1) * META;
                static_for<0, D>([&](auto sdc) { issue_bonds(ptile_at(pt), decltype(sdc)::value, prec); });
                issue_meta(ptile_at(pt), prec);
                issue_ids(ptile_at(pt + 1), pmeta + ((pt + 1) & 1) * META);
            }
#endif
            prio_multiply();
            auto mask_last = [&](f32x4& cur) {           // only the last chunk of a row can be partial or empty
                const int col = 16 * (KC - 1) + 4 * kq;
                if (col >= F) cur.x = 0.f;
                if (col + 1 >= F) cur.y = 0.f;
                if (col + 2 >= F) cur.z = 0.f;
                if (col + 3 >= F) cur.w = 0.f;
            };
            // this wave's share of the partner tile's S1 * KC row pieces: the flat pieces f = NS g + role, g = 0 .. TOTW-1,
            // PW of them after every chunk step (all issued in the first two thirds of the phase, so that the last ones have
            // landed when the phase ends); f -> slot f / KC, chunk f % KC, LDS image 256 f floats into the partner's ring
            constexpr int TOT = S1 * KC, TOTW = (TOT + NS - 1) / NS, STEPS = D * KC;
            constexpr int SPREAD = (2 * STEPS) / 3 > 0 ? (2 * STEPS) / 3 : 1;
            constexpr int PW = (TOTW + SPREAD - 1) / SPREAD;
            auto issue_step = [&](auto stepc) {
                constexpr int step = decltype(stepc)::value;
#ifdef MKGNN_PP_NO_DMA
                return;
#endif
                static_for<0, PW>([&](auto jc) {
                    constexpr int g = step * PW + decltype(jc)::value;
                    if constexpr (g < TOTW) {
                        int f = NS * g + role;
                        if constexpr (NS * g + NS - 1 >= TOT) {                  // (the last round may be short: repeat this wave's previous piece)
                            if (f >= TOT) f -= NS;
                        }
                        // its slot f / KC (wave-uniform, but not a compile-time value: the role is not): a chain of selects, not
                        // an index -- idsD must never be indexed by a run-time value (it would live in scratch)
                        const int slf = f / KC;
                        uint32_t idv = idsD[0];
#pragma unroll
                        for (int q = 1; q < S1; ++q) idv = (slf == q) ? idsD[q] : idv;
                        const int t = f - slf * KC;
                        // a chunk entirely beyond the row's width is fetched from the row's first chunk instead (in bounds,
                        // finite whenever the row is) and masked to zero where it is used
                        const uint32_t off = (t == KC - 1 && 16 * t + 4 * kq >= F) ? idv * xs : idv * xs + 4u * kq + 16u * (uint32_t)t;
                        dma16<SITE_ROWS>(a.x + off, pring + f * 256);
                    }
                });
            };
            // The A-operand reads are inline asm with their own lgkmcnt waits: left to the compiler the read of chunk t + 1,
            // written ahead of chunk t's matrix instructions, is sunk to where its value is first used and every chunk
            // starts with an exposed LDS round trip (the 4-wave kernel's listing shows it too: ds_read_b128, s_waitcnt
            // lgkmcnt(0), v_mfma ...).  Here: wait for step k's registers, issue step k + 1's reads, multiply step k.
            const uint32_t rbase = (uint32_t)(uintptr_t)(ring + lane * 4);      // LDS byte address of this lane's 16 bytes of piece 0
            auto m_phase = [&](auto withc) {
                constexpr bool WC = decltype(withc)::value;           // (HS: the centre belongs to half 0)
                f32x4 bufA[2], bufF[2];
                auto rd = [&](auto stepc) {
                    constexpr int step = decltype(stepc)::value;
                    constexpr bool merged = step >= (D - 1) * KC;
                    constexpr int sl = merged ? D - 1 : step / KC, t = step - sl * KC;
                    lds_read128<(sl * SLOT + t * 256) * 4>(bufA[step & 1], rbase);
                    if constexpr (merged && WC) lds_read128<(D * SLOT + t * 256) * 4>(bufF[step & 1], rbase);
                };
                rd(IC<0>{});
                static_for<0, STEPS>([&](auto stepc) {
                    constexpr int step = decltype(stepc)::value;
                    constexpr bool merged = step >= (D - 1) * KC;
                    constexpr int sl = merged ? D - 1 : step / KC, t = step - sl * KC;
                    if constexpr (merged && WC) lds_wait0(bufA[step & 1], bufF[step & 1]);
                    else lds_wait0(bufA[step & 1]);
                    if constexpr (step + 1 < STEPS) rd(IC<step + 1>{});
                    f32x4 cA = bufA[step & 1];
                    [[maybe_unused]] f32x4 cF = bufF[step & 1];
                    if constexpr (t == KC - 1) { mask_last(cA); if constexpr (merged && WC) mask_last(cF); }
                    if constexpr (BF == 1) {
                        const s16x4s a4 = to_bf16x4s(cA);
#pragma unroll
                        for (int b = 0; b < NBS; ++b) cm[sl][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, bk[b][t], cm[sl][b], 0, 0, 0);
                        if constexpr (merged && WC) cc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(to_bf16x4s(cF), bk[NBS][t], cc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
#pragma unroll
                            for (int b = 0; b < NBS; ++b)
                                cm[sl][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[q4], bk[b][t][q4], cm[sl][b], 0, 0, 0);
                            if constexpr (merged && WC) cc = __builtin_amdgcn_mfma_f32_16x16x4f32(cF[q4], bk[NBS][t][q4], cc, 0, 0, 0);
                        }
                    }
                    issue_step(stepc);
                    if constexpr (HS && step == 2 * KC - 1) __builtin_amdgcn_s_barrier();      // (the partner side's exchange barrier)
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (!HS || half == 0) m_phase(std::true_type{});
            else m_phase(std::false_type{});
            prio_other();
            MKGNN_PHASE(0);
            wait_vmcnt<0>();                             // the partner's next tile has landed (and this wave's stores are out)
            MKGNN_PHASE(1);
            __builtin_amdgcn_s_barrier();                // this team is done reading its ring; the partner may read its own
            MKGNN_PHASE(2);
        } else {
        // (BF == 2) the power-of-two scale of atom ci's row, slot by slot, from the 1 / |x| of the tile's record (landed with slot
        // 0's rows; read behind the compiler's back like the bond rows below).  One register, not S1: the degree-3 body is at
        // the 256-register limit, and a spilled value's reload is a vector-memory operation that drains the DMA queue.
        [[maybe_unused]] const uint32_t sca_addr = (uint32_t)(uintptr_t)(mrec + T::OFF_INV + ci);
        static_for<0, S1>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // ---- multiply slot s: one LDS read per 4 NBS (or 4) matrix instructions, issued one chunk ahead.  (Written as ordinary
            // loads the read of chunk t + 1 -- placed ahead of chunk t's matrix instructions, pinned by sched_group_barrier -- is
            // sunk by the compiler to where its value is first used: the listing shows ds_read_b128, s_waitcnt lgkmcnt(0),
            // v_mfma ... per chunk.  Round 5 measured the alternative, -DMKGNN_FWD_ASM_READS: inline-asm reads with their own
            // lgkmcnt waits, truly a chunk ahead -- 255 VGPRs and the same 65.5 us: the other wave of the SIMD covers that
            // latency.  The ping-pong form, whose multiplying wave has no such partner, keeps the asm reads.)
            prio_multiply();
            [[maybe_unused]] float sca_s = 0.f;
            if constexpr (BF == 2) {
                lds_read32<s * 64>(sca_s, sca_addr);
                lds_wait0(sca_s);
                sca_s = split_row_scale(sca_s);
            }
            if (s < D || !HS || half == 0) {             // (HS: the centre belongs to half 0)
#ifndef MKGNN_FWD_ASM_READS
                const float* rb = ring + buf * SLOT + lane * 4;
                f32x4 nxt = *(const f32x4*)rb;
#else
                const uint32_t rba = (uint32_t)(uintptr_t)(ring + buf * SLOT + lane * 4);     // LDS byte address of this lane's 16 bytes of piece 0
                f32x4 pb[2];
                lds_read128<0>(pb[0], rba);
#endif
                static_for<0, KC>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
#ifndef MKGNN_FWD_ASM_READS
                    f32x4 cur = nxt;
                    if constexpr (t + 1 < KC) nxt = *(const f32x4*)(rb + (t + 1) * 256);
#else
                    lds_wait0(pb[t & 1]);
                    if constexpr (t + 1 < KC) lds_read128<(t + 1) * 1024>(pb[(t + 1) & 1], rba);
                    f32x4 cur = pb[t & 1];
#endif
                    if constexpr (t == KC - 1) {         // only the last chunk of a row can be partial or empty
                        const int col = 16 * t + 4 * kq;
                        if constexpr (PRE) {
                            // the sixteen bytes are hi(0..3) | lo(0..3): element e is half-word e of both halves
                            const uint32_t m01 = (col < F ? 0x0000ffffu : 0u) | (col + 1 < F ? 0xffff0000u : 0u);
                            const uint32_t m23 = (col + 2 < F ? 0x0000ffffu : 0u) | (col + 3 < F ? 0xffff0000u : 0u);
                            cur.x = __uint_as_float(__float_as_uint(cur.x) & m01); cur.z = __uint_as_float(__float_as_uint(cur.z) & m01);
                            cur.y = __uint_as_float(__float_as_uint(cur.y) & m23); cur.w = __uint_as_float(__float_as_uint(cur.w) & m23);
                        } else {
                            if (col >= F) cur.x = 0.f;
                            if (col + 1 >= F) cur.y = 0.f;
                            if (col + 2 >= F) cur.z = 0.f;
                            if (col + 3 >= F) cur.w = 0.f;
                        }
                    }
                    if constexpr (SPL) {
                        SplitReg a4;
                        if constexpr (PRE) a4 = __builtin_bit_cast(SplitReg, cur);       // the producer's split of the same scaled row
                        else a4 = split_scaled(cur, sca_s);
                        if constexpr (s < D) {
#pragma unroll
                            for (int b = 0; b < NBS; ++b) cm[s][b] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4.lo, bk[b][t].hi, cm[s][b], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < NBS; ++b) cm[s][b] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4.hi, bk[b][t].lo, cm[s][b], 0, 0, 0);
#pragma unroll
                            for (int b = 0; b < NBS; ++b) cm[s][b] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4.hi, bk[b][t].hi, cm[s][b], 0, 0, 0);
                        } else {
                            cc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4.lo, bk[NBS][t].hi, cc, 0, 0, 0);
                            cc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4.hi, bk[NBS][t].lo, cc, 0, 0, 0);
                            cc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4.hi, bk[NBS][t].hi, cc, 0, 0, 0);
                        }
                    } else if constexpr (BF == 1) {
                        const s16x4s a4 = to_bf16x4s(cur);
                        if constexpr (s < D) {
#pragma unroll
                            for (int b = 0; b < NBS; ++b) cm[s][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, bk[b][t], cm[s][b], 0, 0, 0);
                        } else {
                            cc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, bk[NBS][t], cc, 0, 0, 0);
                        }
                    } else {
                        if constexpr (s < D) {
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                                for (int b = 0; b < NBS; ++b)
                                    cm[s][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q4], bk[b][t][q4], cm[s][b], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4) cc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[q4], bk[NBS][t][q4], cc, 0, 0, 0);
                        }
#ifndef MKGNN_FWD_ASM_READS
                        // pin the order: the next chunk's read ahead of this chunk's matrix instructions
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, s < D ? 4 * NBS : 4, 0);
#endif
                    }
#ifdef MKGNN_FWD_ASM_READS
                    __builtin_amdgcn_sched_barrier(0);
#endif
                });
            }
            prio_other();
            MKGNN_PHASE(0);
            // ---- retire: the batch of the NEXT step must have landed; the RING - 2 batches issued after it may stay in
            // flight (vmcnt retires in order).  (A 2-buffer ring has just one batch in flight.)
#ifdef MKGNN_EXP_WAIT0                                   // (diagnostic build: every wait drains the DMA queue)
            wait_vmcnt<0>();
#else
            if constexpr (RING == 2) {
                wait_vmcnt<0>();
            } else {
                static_for<0, NS>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    constexpr int n_young = young_batches<D, KC>(s, r);
                    if (role == r) wait_vmcnt<n_young>();
                });
            }
#endif
            MKGNN_PHASE(1);
            if constexpr (NS > 1) __builtin_amdgcn_s_barrier();
            MKGNN_PHASE(2);
            // ---- issue the batch of step k + RING into the buffer just read
            constexpr int sd = (s + RING) % S1;
            const int itd = it + (s + RING) / S1;        // iteration (tile) the DMA pointer is in
            float* const drec = meta + (itd & 1) * META;
            if constexpr (sd == 0) {                     // the DMA pointer enters a new tile: its ids were DMA'd a tile ago
#pragma unroll
                for (int q = 0; q < S1; ++q) idsD[q] = __float_as_uint(drec[16 * q + ci]);
            }
            issue_rows(IC<sd>{}, ring + buf * SLOT);
            if (role == 0) {
                if constexpr (sd < D) issue_bonds(tile_at(itd), sd, drec);
                if constexpr (sd == 0) {
                    issue_meta(tile_at(itd), drec);
                    issue_ids(tile_at(itd + 1), meta + ((itd + 1) & 1) * META);
                }
            }
            buf = (buf + 1 == RING) ? 0 : buf + 1;
            MKGNN_PHASE(3);
        });

        }

        // ---- epilogue: lane = kernel lcol, atoms kq * 4 + jj (the arithmetic of kc_forward_fused).  One atom at a
        // time, fenced: left to itself the scheduler interleaves the atoms' permutation scans for ILP, and with the
        // bank resident in registers that is what spills.
        constexpr int NJ = HS ? 2 : 4;                   // atoms of a lane this wave finishes (HS: 2 * half, 2 * half + 1)
        [[maybe_unused]] float px[NJ][D][2];             // (HS) the partner's two support columns of my atoms
        [[maybe_unused]] float pcc[NJ];
        if constexpr (HS) {
            // swap: my columns of the PARTNER's atoms out, the partner's columns of MY atoms in
            float* const mine = xbuf + (size_t)wave8 * (18 * 64) + lane;
            const float* const theirs = xbuf + (size_t)(wave8 ^ 1) * (18 * 64) + lane;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int s = 0; s < D; ++s)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        mine[((j * D + s) * 2 + b) * 64] = half ? cm[s][b][j] : cm[s][b][2 + j];     // partner's atoms: 2 (1 - half) + j
                mine[(16 + j) * 64] = half ? cc[j] : cc[2 + j];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int s = 0; s < D; ++s)
#pragma unroll
                    for (int b = 0; b < 2; ++b) px[j][s][b] = theirs[((j * D + s) * 2 + b) * 64];
                pcc[j] = theirs[(16 + j) * 64];
            }
        }
        int idx4[NJ];
        float best4[NJ], cen4[NJ];
        [[maybe_unused]] f32x4 inv4[S1];
        if constexpr (!HS) {
#pragma unroll
            for (int s = 0; s < S1; ++s) {
                inv4[s] = *(const f32x4*)(mrec + T::OFF_INV + s * 16 + kq * 4);
                if constexpr (SPL) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) inv4[s][j] = split_inv(inv4[s][j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float m[D][D];
            const int jj = HS ? 2 * half + j : j;        // atom of the lane (run-time only under HS: selects, never an index)
#pragma unroll
            for (int s = 0; s < D; ++s) {
                if constexpr (HS) {
                    float iv = mrec[T::OFF_INV + s * 16 + kq * 4 + jj];
                    if constexpr (SPL) iv = split_inv(iv);
                    const float own0 = half ? cm[s][0][2 + j] : cm[s][0][j], own1 = half ? cm[s][1][2 + j] : cm[s][1][j];
                    m[s][0] = (half ? px[j][s][0] : own0) * iv;
                    m[s][1] = (half ? px[j][s][1] : own1) * iv;
                    m[s][2] = (half ? own0 : px[j][s][0]) * iv;
                    m[s][3] = (half ? own1 : px[j][s][1]) * iv;
                } else {
                    const float iv = inv4[s][j];
#pragma unroll
                    for (int b = 0; b < D; ++b) m[s][b] = cm[s][b][j] * iv;
                }
            }
            best_permutation<D>(m, best4[j], idx4[j]);
            if constexpr (HS) {
                float ivc = mrec[T::OFF_INV + D * 16 + kq * 4 + jj];
                if constexpr (SPL) ivc = split_inv(ivc);
                cen4[j] = (half ? pcc[j] : cc[j]) * ivc;
            }
            else cen4[j] = cc[j] * inv4[D][j];
        }
        MKGNN_PHASE(5);
        uint32_t signw = 0, eqw = 0;
        if constexpr (D == 4) {
            signw = __float_as_uint(mrec[T::OFF_SIGN + kq]);
            eqw = __float_as_uint(mrec[T::OFF_EQ + kq]);
        }
        // unit bond components 2 kq, 2 kq + 1 of (atom ci, slot s).  Read behind the compiler's back: it cannot know that
        // the counted waits above have retired the DMA that wrote them and would drain the whole DMA queue (vmcnt(0))
        // in front of an ordinary LDS read of this part of the record
        float2 eu[D];
        {
            const uint32_t ea = (uint32_t)(uintptr_t)(mrec + T::OFF_BOND + ci * 8 + 2 * kq);      // LDS byte address
            asm volatile("ds_read_b64 %0, %1" : "=v"(eu[0]) : "v"(ea) : "memory");
            if constexpr (D > 1) asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(eu[D > 1 ? 1 : 0]) : "v"(ea) : "memory");
            if constexpr (D > 2) asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(eu[D > 2 ? 2 : 0]) : "v"(ea) : "memory");
            if constexpr (D > 3) asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(eu[D > 3 ? 3 : 0]) : "v"(ea) : "memory");
            if constexpr (D == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eu[0]) : : "memory");
            else if constexpr (D == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eu[0]), "+v"(eu[D > 1 ? 1 : 0]) : : "memory");
            else if constexpr (D == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eu[0]), "+v"(eu[D > 1 ? 1 : 0]), "+v"(eu[D > 2 ? 2 : 0]) : : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(eu[0]), "+v"(eu[D > 1 ? 1 : 0]), "+v"(eu[D > 2 ? 2 : 0]), "+v"(eu[D > 3 ? 3 : 0]) : : "memory");
        }
        // bond-cosine matrices, one (a, b) tile at a time; keep the entry the chosen order uses
        float ed4[NJ][D];
        static_for<0, D>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f32x4 dm[D];                                 // D independent chains
#pragma unroll
            for (int b = 0; b < D; ++b) dm[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(eu[s].x, bv[b].x, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int b = 0; b < D; ++b) dm[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(eu[s].y, bv[b].y, dm[b], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int pb = perm_entry<D, s>(idx4[j]);
                float v;
                if constexpr (HS) {
                    v = half ? dm[0][2 + j] : dm[0][j];
#pragma unroll
                    for (int b = 1; b < D; ++b) v = (pb == b) ? (half ? dm[b][2 + j] : dm[b][j]) : v;
                } else {
                    v = dm[0][j];
#pragma unroll
                    for (int b = 1; b < D; ++b) v = (pb == b) ? dm[b][j] : v;
                }
                ed4[j][s] = v;
            }
        });
        MKGNN_PHASE(6);
        const f32x4 focal4 = *(const f32x4*)(mrec + T::OFF_FOCAL + kq * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int jj = HS ? 2 * half + j : j;
            const int n = tile * 16 + kq * 4 + jj;
            float ed = ed4[j][0];
#pragma unroll
            for (int s = 1; s < D; ++s) ed = __fadd_rn(ed, ed4[j][s]);
            ed = div_by<D>(ed);
#ifdef MKGNN_ABLATE_DIV                                  // (timing experiment)
            float sc = __fadd_rn(__fadd_rn(__fmul_rn(best4[j], ws), __fmul_rn(cen4[j], wc)), __fmul_rn(ed, we)) * wsum;
#else
            float sc = __fadd_rn(__fadd_rn(__fmul_rn(best4[j], ws), __fmul_rn(cen4[j], wc)), __fmul_rn(ed, we)) / wsum;
#endif
            float ch = 1.f;
            if constexpr (D == 4) {
                const int eqb = (int)((eqw >> (8 * jj)) & 0xFFu);
                const float sgn = (float)(int8_t)((signw >> (8 * jj)) & 0xFFu);
                if (do_chir && !eqb) ch = ((float)chirtab[(col_ok ? lcol : 0) * 12 + idx4[j]] == sgn) ? 1.f : -1.f;
                sc *= ch;
            }
            uint32_t focal;
            if constexpr (HS) focal = __float_as_uint(half ? focal4[2 + j] : focal4[j]);
            else focal = __float_as_uint(focal4[j]);
#ifdef MKGNN_ABLATE_STORES                               // (timing experiment: everything computed, nothing stored)
            asm volatile("" :: "v"(sc), "v"(ed), "v"(ch), "v"(focal), "v"(best4[j]), "v"(cen4[j]), "v"(idx4[j]));
            if (false) {
#else
            if (real && col_ok && n < dn) {
#endif
#ifndef MKGNN_ABLATE_OUT                                  // (timing experiments: one kind of store left out)
                a.out[focal * (uint32_t)a.os + (uint32_t)(dg.off + lcol)] = sc;     // (the host checks n_atoms * stride < 2^30)
#else
                asm volatile("" :: "v"(sc), "v"(focal));
#endif
                const uint32_t o = (uint32_t)n * (uint32_t)L + (uint32_t)lcol;   // the host fuses a degree only if N_d * L < 2^32
#ifndef MKGNN_ABLATE_PAIR
#ifndef MKGNN_NO_NT_STORE                                 // streaming stores: the records are written once and read in the backward (forward 45.9 -> 44.5 us)
                if (dg.pair) __builtin_nontemporal_store(f32x4{best4[j], cen4[j], ed, __int_as_float(idx4[j])}, (f32x4*)(dg.pair + 4 * (size_t)o));
#else
                if (dg.pair) pair_store(dg.pair, o, best4[j], cen4[j], ed, idx4[j]);     // one 16-byte record per pair
#endif
#else
                asm volatile("" :: "v"(o), "v"(ed), "v"(best4[j]), "v"(cen4[j]), "v"(idx4[j]));
#endif
                if (dg.chir_out) dg.chir_out[o] = (int8_t)ch;
            }
        }
        MKGNN_PHASE(4);
        if constexpr (PP) __builtin_amdgcn_s_barrier();   // (end of the O phase: nothing to wait for -- the stores may stay in flight)
    }
    if constexpr (PP) {                                  // side 0's share of the phase side 1 sat out at the start
        if (side == 0) {
            __builtin_amdgcn_s_barrier();
            if constexpr (HS) __builtin_amdgcn_s_barrier();
        }
    }
    wait_vmcnt<0>();                                     // no DMA may land in this block's LDS after it is gone
    if (a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * (PP ? 8 : 4) + wave8) * 16;
        o[0] = t_start; o[1] = __builtin_readcyclecounter(); o[2] = (unsigned long long)(D * 16 + cp); o[3] = (unsigned long long)iters;
        o[12] = rt_start; o[13] = __builtin_amdgcn_s_memrealtime();
#ifdef MKGNN_FWD_STAMPS
        for (int i = 0; i < 8; ++i) o[4 + i] = phase[i];
#endif
    }
}

#ifndef MKGNN_EXP_OCC
#define MKGNN_EXP_OCC 2
#endif
#ifndef MKGNN_FWD_PP_DEFAULT
#define MKGNN_FWD_PP_DEFAULT 0
#endif
#ifndef MKGNN_PP_OCC4_KC
#define MKGNN_PP_OCC4_KC 0                               // (experiment: ping-pong bodies up to this chunk count built for 4 waves per SIMD)
#endif
#ifndef MKGNN_FWD_PAIR_DEFAULT
#define MKGNN_FWD_PAIR_DEFAULT 0
#endif
// (KC >= 8, rows of 113 .. 160 floats: the bank alone is up to 160 registers -- one wave per SIMD, 512 registers)
template <int KC, int BF = 0>
__global__ void __launch_bounds__(256, (KC >= 8 ? 1 : MKGNN_EXP_OCC)) kc_forward_stream(FusedFwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int grp = a.blk_group[blockIdx.x];
    const int rank = a.blk_rank[blockIdx.x];
    const int di = a.grp_degree[grp];
    const int cp = a.grp_cp[grp];
    const int count = a.grp_count[grp];
#ifdef MKGNN_EXP_ONLY_D                     // (occupancy experiments: one degree's body only)
    if (di == MKGNN_EXP_ONLY_D - 1) stream_body<MKGNN_EXP_ONLY_D, KC>(a, a.deg[MKGNN_EXP_ONLY_D - 1], cp, rank, count, lds);
    return;
#endif
    switch (di) {
        case 0: stream_body<1, KC, BF>(a, a.deg[0], cp, rank, count, lds); break;
        case 1: stream_body<2, KC, BF>(a, a.deg[1], cp, rank, count, lds); break;
        case 2: stream_body<3, KC, BF>(a, a.deg[2], cp, rank, count, lds); break;
        default: stream_body<4, KC, BF>(a, a.deg[3], cp, rank, count, lds); break;
    }
}

// the ping-pong form: 8 waves, one block per CU, 2 waves per SIMD (KC <= 7: the bank leaves room for two waves)
template <int KC, int BF = 0>
__global__ void __launch_bounds__(512, (KC <= MKGNN_PP_OCC4_KC ? 4 : 2)) kc_forward_pp(FusedFwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int grp = a.blk_group[blockIdx.x];
    const int rank = a.blk_rank[blockIdx.x];
    const int di = a.grp_degree[grp];
    const int cp = a.grp_cp[grp];
    const int count = a.grp_count[grp];
    switch (di) {
        case 0: stream_body<1, KC, BF, true>(a, a.deg[0], cp, rank, count, lds); break;
        case 1: stream_body<2, KC, BF, true>(a, a.deg[1], cp, rank, count, lds); break;
        case 2: stream_body<3, KC, BF, true>(a, a.deg[2], cp, rank, count, lds); break;
        default: stream_body<4, KC, BF, true>(a, a.deg[3], cp, rank, count, lds); break;
    }
}

// ---------------------------------------------------------------- host ----
// the bf16 similarity variant runs on the streamed kernel for the model's own row widths (KC = 2: F <= 32; KC = 7: 97 .. 112)
bool stream_forward_bf16_supported(int F) { const int KC = (F + 15) / 16; return KC == 2 || KC == 7; }

bool stream_forward_supported(int d, int F, int E, int L, int64_t n_atoms, int64_t x_stride, int64_t out_stride, const float* e_unit) {
    if (d < 1 || d > 4 || L < 1 || E < 1 || E > 8 || !e_unit) return false;
    if (F < 1 || F > STREAM_MAX_F || !bank_pitch(F)) return false;   // KC = ceil(F / 16) <= 10 chunks; only the last may be partial
    if (d == 4 && L * 12 > 768) return false;            // the chirality sign table's LDS slot
    if (stream_column_parts(d, L) > 8) return false;     // (a bank of > 128 / 256 kernels of one degree: the generic kernels)
    // 32-bit element offsets into x, out and the unit bond rows
    if ((uint64_t)n_atoms * (uint64_t)x_stride >= (1ull << 30) || (uint64_t)n_atoms * (uint64_t)out_stride >= (1ull << 30) ||
        (uint64_t)n_atoms * 32ull >= (1ull << 30))
        return false;
    return true;
}

__global__ void unit_rows8_kernel(const float* __restrict__ in, int64_t n, int E, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = k < E ? in[r * E + k] : 0.f;
    // the summation order of the forward kernels' in-register form: pairs (2k, 2k+1), then (p0 + p1) + (p2 + p3)
    float p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = fmaf(e[2 * k + 1], e[2 * k + 1], e[2 * k] * e[2 * k]);
    const float s2 = __fadd_rn(__fadd_rn(p[0], p[1]), __fadd_rn(p[2], p[3]));
    const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
    f32x4 lo = {e[0] * ie, e[1] * ie, e[2] * ie, e[3] * ie}, hi = {e[4] * ie, e[5] * ie, e[6] * ie, e[7] * ie};
    *(f32x4*)(out + r * 8) = lo;
    *(f32x4*)(out + r * 8 + 4) = hi;
}

hipError_t launch_unit_rows8(const float* in, int64_t n_rows, int E, float* out, hipStream_t st) {
    if (n_rows == 0) return hipSuccess;
    unit_rows8_kernel<<<(unsigned)((n_rows + 255) / 256), 256, 0, st>>>(in, n_rows, E, out);
    return hipGetLastError();
}

// Block table of the streamed launch: one group per degree, block counts by greedy min-max over
// prologue + iterations * cost per tile (all blocks are resident at once: the launch lasts as long as its slowest wave),
// groups interleaved over the block ids, the blocks of a group that share an XCD (block id mod 8) given adjacent runs
// of tiles (the buckets are sorted by atom id: what one group gathers as neighbours another gathers as focal rows).
static size_t plan_stream(FusedFwdArgs& a, const bool use[4], int KC, int* nblocks_out, bool pp = false, bool split = false) {
    constexpr int MG = FUSED_MAX_GROUPS;                // groups = (degree, column part); launch_forward_stream checks the total
    double cost[MG];
    int64_t tiles_of[MG], cap[MG];
    int nstream_of[MG], ng = 0;
    size_t lds_floats = 0;
    for (int i = 0; i < 4; ++i) {
        if (!use[i]) continue;
        FusedDeg& g = a.deg[i];
        const int d = i + 1;
        g.nct = (g.L + 15) / 16;
        g.kpt = (g.L + g.nct - 1) / g.nct;
        g.cs = stream_column_parts(d, g.L); g.nloc = stream_tiles_per_part(d); g.ics = g.nloc;
        // streams per block: NSTREAM of the degree's body (a ping-pong block is two 4-wave teams)
        const int nstream = ((d == 1) ? 4 : (d == 4 ? 1 : 2)) * (pp ? 2 : 1);
        const int64_t ntiles = (g.n + 15) / 16;
        const size_t fl = (size_t)(pp ? stream_pp_lds_floats(d, KC) : stream_lds_floats(d, KC));
        if (fl > lds_floats) lds_floats = fl;
        for (int cp = 0; cp < g.cs; ++cp) {
            // what a wave takes per tile, everything included (multiply, DMA issue, waits, epilogue), in units of 32
            // cycles -- measured with tools/stream_stamps.py at two waves per SIMD and all groups resident (F = 110, batch
            // 4096: 10.2 k / 15.2 k / 27.3 k / 33.8 k cycles for degree 1..4), scaled with the chunk count for F <= 32
            // (other chunk counts: interpolated)
            static double calib7[4] = {319.0, 475.0, 853.0, 1056.0}, calib2[4] = {200.0, 260.0, 420.0, 560.0};
            // (the split-fp16 products, round 5: 6.3 k / 10.8 k / 17.0 k / 22.7 k cycles per tile at F = 110, 4.2 k / 5.6 k / 9.7 k /
            // 14.8 k at F = 28)
            static double split7[4] = {230.0, 337.0, 533.0, 710.0}, split2[4] = {131.0, 176.0, 303.0, 462.0};
            static const bool env_read = [] {            // diagnostics: MKGNN_STREAM_COST="c1,c2,c3,c4" (applies to both widths)
                if (const char* e = getenv("MKGNN_STREAM_COST")) {
                    double v[4];
                    if (sscanf(e, "%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3]) == 4)
                        for (int k = 0; k < 4; ++k) calib7[k] = calib2[k] = split7[k] = split2[k] = v[k];
                }
                return true;
            }();
            (void)env_read;
            if (ng >= MG) { *nblocks_out = -1; return 0; }
            cost[ng] = split ? split2[i] + (split7[i] - split2[i]) * (KC - 2) / 5.0 : calib2[i] + (calib7[i] - calib2[i]) * (KC - 2) / 5.0;
            tiles_of[ng] = ntiles;
            cap[ng] = (ntiles + nstream - 1) / nstream;
            nstream_of[ng] = nstream;
            a.grp_degree[ng] = (uint8_t)i;
            a.grp_cp[ng] = (uint8_t)cp;
            ++ng;
        }
    }
    if (ng == 0) { *nblocks_out = 0; return 0; }
    const double prologue = 60.0;
    auto finish = [&](int g, int blocks) {
        const int64_t streams = (int64_t)blocks * nstream_of[g];
        return prologue + (double)((tiles_of[g] + streams - 1) / streams) * cost[g];
    };
    int count[MG], nb = 0;
    for (int g = 0; g < ng; ++g) { count[g] = 1; ++nb; }
    // (a ping-pong block is 8 waves, one per CU: half as many blocks for the same number of waves, run-time caps included)
    // (... unless the ping-pong body of this chunk count is built for two blocks per CU: MKGNN_PP_OCC4_KC)
    const int max_blocks = (pp && KC > MKGNN_PP_OCC4_KC) ? (grid_cap(g_grid_caps.fwd, FUSED_MAX_BLOCKS) + 1) / 2
                                                         : grid_cap(g_grid_caps.fwd, FUSED_MAX_BLOCKS);
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
    // Which blocks share a CU matters: two co-resident blocks of DIFFERENT degrees run at different speeds than two of the
    // same degree, and the launch lasts as long as its unluckiest wave (round 5 stamps: the waves of one group, all with the
    // same number of tiles, ended between 52 and 69 us).  MKGNN_FWD_PAIR = P (diagnostics; 0 = the interleave of rounds 1-4):
    // blocks b and b + P get the same group where the counts allow it.
    static const int pair_dist = [] { const char* e = getenv("MKGNN_FWD_PAIR"); return e ? atoi(e) : MKGNN_FWD_PAIR_DEFAULT; }();
    int given[MG] = {};
    if (!pp && pair_dist > 0 && nb == 2 * 256 && (pair_dist == 256 || pair_dist == 8)) {
        // deal PAIRS (b, b + P): half of every group's blocks to the first member, the same groups to the second; a group
        // with an odd count shares one CU with another odd one
        int half_cnt[MG], odd[MG], nodd = 0;
        for (int g = 0; g < ng; ++g) { half_cnt[g] = count[g] / 2; if (count[g] & 1) odd[nodd++] = g; }
        int seq[256], npairs = 0, hg[MG] = {};
        int full_pairs = 0;
        for (int g = 0; g < ng; ++g) full_pairs += half_cnt[g];
        for (int k = 0; k < full_pairs; ++k) {
            int pick = -1;
            double best = -1e30;
            for (int g = 0; g < ng; ++g) {
                if (hg[g] >= half_cnt[g]) continue;
                const double lag = (double)half_cnt[g] * (k + 1) / full_pairs - hg[g];
                if (lag > best) { best = lag; pick = g; }
            }
            seq[npairs++] = pick; ++hg[pick];
        }
        // first members: pair slots in order; odd leftovers fill the remaining slots two by two
        int first[256], second[256];
        for (int k = 0; k < npairs; ++k) first[k] = second[k] = seq[k];
        for (int k = 0; k + 1 < nodd + 1 && npairs < 256; k += 2) {
            first[npairs] = odd[k]; second[npairs] = (k + 1 < nodd) ? odd[k + 1] : odd[k]; ++npairs;
        }
        if (npairs == 256) {
            for (int k = 0; k < 256; ++k) {
                int b0, b1;
                if (pair_dist == 256) { b0 = k; b1 = k + 256; }
                else { b0 = (k / 8) * 16 + (k % 8); b1 = b0 + 8; }       // (b, b + 8): neighbours on one XCD
                a.blk_group[b0] = (uint8_t)first[k]; a.blk_group[b1] = (uint8_t)second[k];
            }
            for (int b = 0; b < nb; ++b) ++given[a.blk_group[b]];
            for (int g = 0; g < ng; ++g) if (given[g] != count[g]) given[0] = -1;      // (cannot happen; falls through to the interleave)
        } else given[0] = -1;
    } else given[0] = -1;
    if (given[0] < 0) {
        for (int g = 0; g < ng; ++g) given[g] = 0;
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
    for (int g = 0; g < ng; ++g) a.grp_count[g] = (uint16_t)count[g];
    note_plan(0, nb, ng, tiles_of, count, nstream_of);
    *nblocks_out = nb;
    return lds_floats * 4;
}

static unsigned long long* g_stream_stamps = nullptr;
extern "C" int mkgnn_debug_set_stream_stamps(void* device_ptr) { g_stream_stamps = (unsigned long long*)device_ptr; return 0; }

// groups (degree, column part) a streamed launch over these degrees needs (budget: FUSED_MAX_GROUPS)
int stream_forward_groups(const int L[4], const bool use[4]) {
    int n = 0;
    for (int i = 0; i < 4; ++i) if (use[i]) n += stream_column_parts(i + 1, L[i]);
    return n;
}

template <int KC, int BF = 0> static hipError_t launch_stream_kc(const FusedFwdArgs& a, int nb, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 64 * 1024) {                         // (two such blocks still fit a CU's 160 KB; KC >= 8: one block per CU)
        static PerDeviceOnce attr_set;
        if (const int slot = attr_set.pending(); slot >= 0) {
            hipError_t e = hipFuncSetAttribute((const void*)kc_forward_stream<KC, BF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (KC >= 8 ? 128 : 80) * 1024);
            if (e != hipSuccess) return e;
            attr_set.set(slot);
        }
    }
    kc_forward_stream<KC, BF><<<nb, 256, lds_bytes, st>>>(a);
    return hipGetLastError();
}

template <int KC, int BF = 0> static hipError_t launch_pp_kc(const FusedFwdArgs& a, int nb, size_t lds_bytes, hipStream_t st) {
    static PerDeviceOnce attr_set;
    if (const int slot = attr_set.pending(); slot >= 0) {
        hipError_t e = hipFuncSetAttribute((const void*)kc_forward_pp<KC, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.set(slot);
    }
    kc_forward_pp<KC, BF><<<nb, 512, lds_bytes, st>>>(a);
    return hipGetLastError();
}

// MKGNN_FWD_PP: 0 = the 4-wave blocks of rounds 2-4, 1 = ping-pong blocks wherever they apply (KC <= 7), unset = default
static int fwd_pp_mode() {
    static const int m = [] { const char* e = getenv("MKGNN_FWD_PP"); return e ? atoi(e) : MKGNN_FWD_PP_DEFAULT; }();
    return m;
}

// MKGNN_FWD_SPLIT: 1 = the node-feature products as split fp16 (BF = 2), 0 = v_mfma_f32_16x16x4_f32, unset = default
#ifndef MKGNN_FWD_SPLIT_DEFAULT
#define MKGNN_FWD_SPLIT_DEFAULT 1
#endif
static std::atomic<int> g_fwd_split_override{-1};
static int fwd_split_mode() {
    static const int m = [] { const char* e = getenv("MKGNN_FWD_SPLIT"); return e ? atoi(e) : MKGNN_FWD_SPLIT_DEFAULT; }();
    const int o = g_fwd_split_override.load(std::memory_order_relaxed);
    return o >= 0 ? o : m;
}
// tests: 1 / 0 = split-fp16 / fp32 matrix instructions from the next launch on, -1 = what the environment says
extern "C" int mkgnn_debug_set_forward_products(int32_t mode) { g_fwd_split_override.store(mode < 0 ? -1 : (mode ? 1 : 0)); return 0; }

// pre-split atom rows (kgnn_split.h): the streamed kernel with split-fp16 products is the only forward that reads them
bool stream_rows_split_supported(int F) { return (F + 15) / 16 <= 7 && fwd_split_mode() != 0; }

hipError_t launch_forward_stream(FusedFwdArgs& a, const bool use[4], hipStream_t st) {
    const int KC = (a.F + 15) / 16;
    a.stamps = g_stream_stamps;
    a.FPB = bank_pitch(a.F);
    int nb = 0;
    if (KC <= 7 && fwd_pp_mode() != 0 && !a.x_split) {
        const size_t lds_pp = plan_stream(a, use, KC, &nb, true);
        if (nb == 0) return hipSuccess;
        if (nb > 0 && lds_pp <= (size_t)160 * 1024) {
            g_last_plan[0].launches.fetch_add(1);
            if (a.bf16) {
                if (KC == 2) return launch_pp_kc<2, 1>(a, nb, lds_pp, st);
                if (KC == 7) return launch_pp_kc<7, 1>(a, nb, lds_pp, st);
                return hipErrorInvalidValue;
            }
            switch (KC) {
                case 1: return launch_pp_kc<1>(a, nb, lds_pp, st);
                case 2: return launch_pp_kc<2>(a, nb, lds_pp, st);
                case 3: return launch_pp_kc<3>(a, nb, lds_pp, st);
                case 4: return launch_pp_kc<4>(a, nb, lds_pp, st);
                case 5: return launch_pp_kc<5>(a, nb, lds_pp, st);
                case 6: return launch_pp_kc<6>(a, nb, lds_pp, st);
                default: return launch_pp_kc<7>(a, nb, lds_pp, st);
            }
        }
    }
    // (rows of more than 112 floats, KC >= 8, keep the fp32 matrix instructions: one wave per SIMD there, a rare shape)
    const bool split = !a.bf16 && KC <= 7 && fwd_split_mode() != 0;
    const size_t lds_bytes = plan_stream(a, use, KC, &nb, false, split);
    if (nb == 0) return hipSuccess;
    if (nb < 0 || lds_bytes > (size_t)(KC >= 8 ? 128 : 80) * 1024) return hipErrorInvalidValue;     // (the caller checks stream_forward_groups first)
    g_last_plan[0].launches.fetch_add(1);
    if (a.bf16) {
        if (KC == 2) return launch_stream_kc<2, 1>(a, nb, lds_bytes, st);
        if (KC == 7) return launch_stream_kc<7, 1>(a, nb, lds_bytes, st);
        return hipErrorInvalidValue;                     // (the caller asks stream_forward_bf16_supported first)
    }
    if (a.x_split) {                                     // (the caller has asked stream_rows_split_supported)
        if (!split) return hipErrorInvalidValue;
        switch (KC) {
            case 1: return launch_stream_kc<1, 3>(a, nb, lds_bytes, st);
            case 2: return launch_stream_kc<2, 3>(a, nb, lds_bytes, st);
            case 3: return launch_stream_kc<3, 3>(a, nb, lds_bytes, st);
            case 4: return launch_stream_kc<4, 3>(a, nb, lds_bytes, st);
            case 5: return launch_stream_kc<5, 3>(a, nb, lds_bytes, st);
            case 6: return launch_stream_kc<6, 3>(a, nb, lds_bytes, st);
            default: return launch_stream_kc<7, 3>(a, nb, lds_bytes, st);
        }
    }
    if (split) {
        switch (KC) {
            case 1: return launch_stream_kc<1, 2>(a, nb, lds_bytes, st);
            case 2: return launch_stream_kc<2, 2>(a, nb, lds_bytes, st);
            case 3: return launch_stream_kc<3, 2>(a, nb, lds_bytes, st);
            case 4: return launch_stream_kc<4, 2>(a, nb, lds_bytes, st);
            case 5: return launch_stream_kc<5, 2>(a, nb, lds_bytes, st);
            case 6: return launch_stream_kc<6, 2>(a, nb, lds_bytes, st);
            default: return launch_stream_kc<7, 2>(a, nb, lds_bytes, st);
        }
    }
    switch (KC) {
        case 1: return launch_stream_kc<1>(a, nb, lds_bytes, st);
        case 2: return launch_stream_kc<2>(a, nb, lds_bytes, st);
        case 3: return launch_stream_kc<3>(a, nb, lds_bytes, st);
        case 4: return launch_stream_kc<4>(a, nb, lds_bytes, st);
        case 5: return launch_stream_kc<5>(a, nb, lds_bytes, st);
        case 6: return launch_stream_kc<6>(a, nb, lds_bytes, st);
        case 7: return launch_stream_kc<7>(a, nb, lds_bytes, st);
        case 8: return launch_stream_kc<8>(a, nb, lds_bytes, st);
        case 9: return launch_stream_kc<9>(a, nb, lds_bytes, st);
        case 10: return launch_stream_kc<10>(a, nb, lds_bytes, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mkgnn
