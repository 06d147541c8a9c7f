// CSR row gather-sums on 16-byte aligned rows: the three HBM-bound passes around the kernel convolution.
//
//   segment_sum   out[i] = sum_k in[col[k]]            MolGCN.propagate, aggr='add' (KernelLayer.py:119-123)
//                 (+ 1 / max(|out[i]|, eps) for the next layer's cosine)
//   gather        gx[j]  = undo-normalisation( sum_k contrib[rows[k]] )   backward of the index_select
//                 gathers (kernels.py:527, 543) and of the row normalisation inside the cosine
//   row_inv_norm  1 / max(|x[i]|, eps)
//
// These passes are latency bound, not bandwidth bound, when written one row per wave with the
// rowptr -> index -> row dependency chain exposed (three round trips per row: measured 48 us for 140 MB).
// Here a row of W <= 256 floats is held by LPR = 8/16/32/64 lanes (one float4 each), a wave works on
// 64 / LPR rows at once, and the chain is software pipelined two deep: while the rows of group g are
// in flight, the indices of group g + 1 and the row pointers of group g + 2 are fetched, so one group
// costs one round trip.  All loads are unconditional (clamped addresses, masked values): a load under
// a lane-dependent branch ends the basic block with a full s_waitcnt and serialises the pipeline.
// Sums run in CSR order and the row reductions are fixed xor trees: results are reproducible.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include "kgnn_common.h"
#include "kgnn_launch.h"
#include "kgnn_split.h"

#include <type_traits>

namespace mkgnn {

typedef mkgnn_f32x4 f32x4;

struct CsrArgs {
    const float* src; int64_t ss;          // gathered rows
    const int32_t* rowptr; const int32_t* idx;
    int64_t n; int width;
    float* out; int64_t os;
    float* inv_out;                        // segment sum: optional row norms of out
    const float* x; int64_t xs; const float* inv;   // gather: the forward's rows and their norms
    // block rows (BLK != 0): a row of a kernel-convolution output is non-zero only in the column block of its atom's
    // degree d: columns [off(d), off(d) + len(d)), byte d of the two packed tables (len 0 = no block)
    const int8_t* deg8;                    // BLK == 2: degree of every destination row
    uint64_t blk_off, blk_len;
    int fixed4;                            // diagnostics (MKGNN_CSR_FIXED4=1): four row loads per group whatever the segments hold
    // round 6, pre-split rows (kgnn_split.h, split_row_store): a row kept as the fp16 halves hi | lo of x * 2^(exponent(1 / |x|) + 8),
    // sixteen bytes per four floats -- exactly what the streamed kernels' matrix instructions take
    int split_out;                         // segment sum with inv_out: the written rows are pre-split
    int x_split;                           // gather: the forward's rows x are pre-split
};

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// v where `keep`, else +0 -- as a bit mask, not a branch or a select on the value: the compiler turns "if (c) acc += v"
// on a freshly loaded v into a conditional block that CONTAINS the load and ends with s_waitcnt vmcnt(0), which waits
// for every load in flight (the prefetched indices and row pointers of the next groups included).
__device__ __forceinline__ f32x4 keep_if(f32x4 v, bool keep) {
    const uint32_t m = keep ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = __uint_as_float(__float_as_uint(v[c]) & m);
    return v;
}

__device__ __forceinline__ f32x4 mask_cols(f32x4 v, int col, int width) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (col + c < width) ? v[c] : 0.f;
    return v;
}

// BLK = 1: the gathered rows are block rows -- an index entry carries its row's degree in bits 28..30, a lane loads
//          only where its four columns meet that block and keeps only the block's columns (what lies outside the
//          block in memory is never used: the producer need not zero it);
// BLK = 2: the written rows are block rows -- only the columns of the destination's own block are summed and stored.
// The loads stay unconditional: see row_load.
template <int LPR, bool GATHER, int BLK = 0>
__global__ void __launch_bounds__(256) csr_rows_kernel(CsrArgs a) {
    static_assert(!(GATHER && BLK), "block rows: segment sums only");
    constexpr int RPW = 64 / LPR, SEG = 4;
    static_assert(LPR >= SEG, "one index per lane of a row group");
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, l = lane % LPR;
    const int col = 4 * l;
    const bool active = col < a.width;                  // this lane holds columns col .. col + 3
    const int colc = active ? col : 0;
    // 32-bit row / group arithmetic throughout (the host launches these kernels for n < 2^31 - 2^20 rows; rowptr is
    // int32 anyway): the 64-bit compares and multiply-adds of the index bookkeeping were a third of the VALU work,
    // and the VALU is ~50 % busy in these kernels (PMC: 436 VALU instructions per wave for 3 groups of 2 rows).
    const int n = (int)a.n;
    const int wave0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const int ngroups = (n + RPW - 1) / RPW;
    int last = a.rowptr[n] - 1;
    const int32_t* idxp = last >= 0 ? a.idx : a.rowptr;  // an empty index list is never dereferenced
    last = last >= 0 ? last : 0;

    auto segment = [&](int g, int& lo, int& hi, int& dg) -> int {
        const int j = g * RPW + sub;
        const int jc = (j < n && g < ngroups) ? j : n - 1;
        lo = a.rowptr[(uint32_t)jc];
        hi = a.rowptr[(uint32_t)jc + 1];
        if constexpr (BLK == 2) dg = a.deg8[(uint32_t)jc];
        if (j >= n || g >= ngroups) hi = lo;
        return jc;
    };
    // [b0, b1) = column block of degree d
    auto block_of = [&](int d, int& b0, int& b1) {
        b0 = (int)((a.blk_off >> (8 * (d & 7))) & 0xFF);
        b1 = b0 + (int)((a.blk_len >> (8 * (d & 7))) & 0xFF);
    };
    // BLK == 1: this lane's columns are fixed, so what it does with a source of degree d is too: which of its four
    // elements lie in block d (4 bits) and which 16 bytes it reads (its own, or the nearest of the block).  Both
    // tables fit one register each (degrees 0..4, 4 resp. 6 bits per entry).
    uint32_t keep_tab = 0, col_tab = 0;
    if constexpr (BLK == 1) {
#pragma unroll
        for (int d = 1; d <= 4; ++d) {
            int b0, b1;
            block_of(d, b0, b1);
            uint32_t m = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) m |= (uint32_t)(col + c >= b0 && col + c < b1) << c;
            const int lo4 = b0 & ~3, hi4 = (b1 - 1) & ~3;
            int cc = colc < lo4 ? lo4 : (colc > hi4 ? hi4 : colc);
            if (b1 <= b0) cc = 0;
            keep_tab |= m << (4 * d);
            col_tab |= (uint32_t)(cc >> 2) << (6 * d);
        }
    }
    // A lane whose four columns miss the block reads the nearest 16 bytes of the block instead -- the same cache
    // lines the other lanes of its row fetch, so it adds no traffic (a fixed dummy address made one hot spot of a
    // few lines that every wave hammered) -- and its value is masked / never stored.
    auto row_load = [&](int rid_raw, int dst_b0, int dst_b1) -> f32x4 {
        int row = rid_raw, c = colc;
        if constexpr (BLK == 1) {
            const uint32_t d = (uint32_t)rid_raw >> 28;
            row = rid_raw & 0x0FFFFFFF;
            c = (int)((col_tab >> (6 * d)) & 63) << 2;
        } else if constexpr (BLK == 2) {
            const int lo4 = dst_b0 & ~3, hi4 = (dst_b1 - 1) & ~3;
            c = c < lo4 ? lo4 : (c > hi4 ? hi4 : c);
            if (dst_b1 <= dst_b0) c = 0;                  // no block at all (atom in no bucket): any valid address
        }
#ifdef MKGNN_CSR_NT_LOAD                                  // (A/B build, not kept: 43.5 -> 44.6 us alone, the step unchanged)
        if constexpr (GATHER) return __builtin_nontemporal_load((const f32x4*)(a.src + (uint64_t)(uint32_t)row * (uint32_t)a.ss + (uint32_t)c));
#endif
        return *(const f32x4*)(a.src + (uint64_t)(uint32_t)row * (uint32_t)a.ss + (uint32_t)c);      // one v_mad_u64_u32
    };
    // v where the source is valid and, BLK == 1, inside the source's own block; +0 elsewhere (bit masks: see keep_if)
    auto row_keep = [&](f32x4 v, int rid_raw, bool valid) -> f32x4 {
        if constexpr (BLK == 1) {
            const int m = valid ? (int)(keep_tab >> (4 * ((uint32_t)rid_raw >> 28))) : 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                v[c] = __uint_as_float(__float_as_uint(v[c]) & (uint32_t)__builtin_amdgcn_sbfe(m, c, 1));   // bit c -> 0 / ~0
            return v;
        } else {
            return keep_if(v, valid);
        }
    };
    // The SEG indices of a row: ONE load instruction per wave -- lane l of the row's group fetches entry
    // k0 + (l % SEG) -- and a crossbar broadcast (ds_bpermute, no memory pipe) when they are used, one group later.
    // Every lane loading its own copy of all SEG entries cost SEG address-pipe passes per group for two distinct
    // cache lines; the address pipe, not HBM, bounds these kernels (the block-row forms, with a fifth of the bytes,
    // take the same time).
    auto idx_issue = [&](int lo, int hi, int k0) -> int {
        const int u_mine = l & (SEG - 1);
        int k = k0 + u_mine < hi ? k0 + u_mine : lo;
        k = k < last ? k : last;
        return idxp[(uint32_t)k];
    };
    auto idx_bcast = [&](int mine, int (&rid)[SEG]) {
#pragma unroll
        for (int u = 0; u < SEG; ++u) rid[u] = __shfl(mine, sub * LPR + u, 64);
    };

    int lo_c, hi_c, lo_n, hi_n, dg_c = 0, dg_n = 0;
    int g = wave0;
    int j_c = segment(g, lo_c, hi_c, dg_c);
    int mine_c = idx_issue(lo_c, hi_c, lo_c);
    int j_n = segment(g + nwaves, lo_n, hi_n, dg_n);
    // The loop body for K = 1 .. SEG row loads per group (round 5).  A group used to issue SEG = 4 row loads whatever its rows'
    // segments held -- 2.1 sources per atom on average in `propagate`, 3.1 roles in the gather -- clamped and masked; these
    // passes are bound by the address pipe and the VALU, not by bytes, so the loads that fetch nothing cost what
    // the others cost.  K = the longest segment among the wave's rows (wave-uniform, from the row pointers that arrived a
    // group ago), every arm a complete body: the loads in flight across the loop's back edge are the same in every arm, so
    // the compiler's wait counts stay static.  The sums are the same sums (the dropped terms were +0).
    int mine_n, lo_nn, hi_nn, dg_nn, j_nn;
    auto body = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        dg_nn = 0;
        int d0 = 0, d1 = 0;                                // BLK == 2: the destination's block
        if constexpr (BLK == 2) block_of(dg_c, d0, d1);
        int rid_c[SEG];
        idx_bcast(mine_c, rid_c);
        f32x4 v[SEG];
#pragma unroll
        for (int u = 0; u < K; ++u) v[u] = row_load(rid_c[u], d0, d1);
        f32x4 xv;
        float iv = 0.f;
        if constexpr (GATHER) {
            xv = *(const f32x4*)(a.x + (uint64_t)(uint32_t)j_c * (uint32_t)a.xs + colc);
            iv = a.inv[j_c];
        }
        mine_n = idx_issue(lo_n, hi_n, lo_n);
        j_nn = segment(g + 2 * nwaves, lo_nn, hi_nn, dg_nn);
        // ---- consume
        f32x4 acc = row_keep(v[0], rid_c[0], lo_c < hi_c);
#pragma unroll
        for (int u = 1; u < K; ++u) acc += row_keep(v[u], rid_c[u], lo_c + u < hi_c);
        if (K == SEG && __any(hi_c - lo_c > SEG)) {                 // long segments: rare (more than four bonds / five roles)
            for (int k0 = lo_c + SEG; __any(k0 < hi_c); k0 += SEG) {
                int rid[SEG];
                idx_bcast(idx_issue(lo_c, hi_c, k0), rid);
                f32x4 w[SEG];
#pragma unroll
                for (int u = 0; u < SEG; ++u) w[u] = row_load(rid[u], d0, d1);
#pragma unroll
                for (int u = 0; u < SEG; ++u) acc += row_keep(w[u], rid[u], k0 + u < hi_c);
            }
        }
        acc = mask_cols(acc, active ? col : a.width, a.width);
        const bool row_ok = g * RPW + sub < n;
        if constexpr (GATHER) {
            // d/dx of x / max(|x|, eps): (acc - (acc . xh) xh) * inv, or acc * inv where the clamp is active
            // (pre-split rows: hi + lo is the scaled row to 2^-22 of an element; the power of two leaves through 1 / |x|)
            if (a.x_split) { xv = split_row_value(xv); }
            f32x4 xh = mask_cols(xv, active ? col : a.width, a.width) * (a.x_split ? split_row_inv(iv) : iv);
            float dotp = acc[0] * xh[0];
            dotp = fmaf(acc[1], xh[1], dotp); dotp = fmaf(acc[2], xh[2], dotp); dotp = fmaf(acc[3], xh[3], dotp);
            dotp = group_sum<LPR>(dotp);
            const bool clamped = iv >= (1.f / MKGNN_EPS);
            f32x4 r;
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = clamped ? acc[c] * iv : (acc[c] - dotp * xh[c]) * iv;
            if (row_ok && active) *(f32x4*)(a.out + (uint64_t)(uint32_t)j_c * (uint32_t)a.os + col) = r;
        } else if constexpr (BLK == 2) {
            // only the destination's own block is defined; the rest of its row is left as it is
            if (row_ok && active && col + 3 >= d0 && col < d1) {
                float* dst = a.out + (uint64_t)(uint32_t)j_c * (uint32_t)a.os + col;
                if (col >= d0 && col + 3 < d1) *(f32x4*)dst = acc;
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (col + c >= d0 && col + c < d1) dst[c] = acc[c];
                }
            }
        } else {
            f32x4 stored = acc;                                     // alignment padding is written as zero
            if (a.inv_out) {
                float ss = acc[0] * acc[0];
                ss = fmaf(acc[1], acc[1], ss); ss = fmaf(acc[2], acc[2], ss); ss = fmaf(acc[3], acc[3], ss);
                ss = group_sum<LPR>(ss);
                const float inv = 1.f / fmaxf(sqrtf(ss), MKGNN_EPS);
                if (row_ok && l == 0) a.inv_out[j_c] = inv;
                if (a.split_out) stored = split_row_store(acc, inv);
            }
            if (row_ok && active) *(f32x4*)(a.out + (uint64_t)(uint32_t)j_c * (uint32_t)a.os + col) = stored;
        }
    };
    for (; g < ngroups; g += nwaves) {
        const int len = hi_c - lo_c;
        const int need = a.fixed4 ? SEG : 1 + (int)__any(len >= 2) + (int)__any(len >= 3) + (int)__any(len >= 4);
        switch (need) {
            case 1: body(std::integral_constant<int, 1>{}); break;
            case 2: body(std::integral_constant<int, 2>{}); break;
            case 3: body(std::integral_constant<int, 3>{}); break;
            default: body(std::integral_constant<int, SEG>{}); break;
        }
        // ---- shift the pipeline
        lo_c = lo_n; hi_c = hi_n; j_c = j_n; dg_c = dg_n; mine_c = mine_n;
        lo_n = lo_nn; hi_n = hi_nn; j_n = j_nn; dg_n = dg_nn;
    }
}

// same lane assignment and reduction order as csr_rows_kernel's fused norm: bit-identical results
template <int LPR>
__global__ void __launch_bounds__(256) row_inv_norm_aligned_kernel(const float* __restrict__ x, int64_t xs, int64_t n, int width,
                                                                   float* __restrict__ inv) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, l = lane % LPR, col = 4 * l;
    const bool active = col < width;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ngroups = (n + RPW - 1) / RPW;
    for (int64_t g = wave0; g < ngroups; g += 2 * nwaves) {
        f32x4 v[2];
        int64_t j[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            j[u] = (g + u * nwaves) * RPW + sub;
            const int64_t jc = j[u] < n ? j[u] : n - 1;
            v[u] = *(const f32x4*)(x + jc * xs + (active ? col : 0));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 m = mask_cols(v[u], active ? col : width, width);
            float ss = m[0] * m[0];
            ss = fmaf(m[1], m[1], ss); ss = fmaf(m[2], m[2], ss); ss = fmaf(m[3], m[3], ss);
            ss = group_sum<LPR>(ss);
            if (j[u] < n && l == 0) inv[j[u]] = 1.f / fmaxf(sqrtf(ss), MKGNN_EPS);
        }
    }
}

// 1 / max(|x[i]|, eps) AND the row itself rewritten as pre-split rows (kgnn_split.h): what a producer inside the library does in
// its own epilogue (csr_rows_kernel, bn_apply_kernel), for rows that come from outside -- benchmarks and tests that feed the
// streamed kernels the operand form the training step feeds them (mkgnn_rows_presplit).  Same lanes, same sums as above.
template <int LPR>
__global__ void __launch_bounds__(256) row_presplit_kernel(const float* __restrict__ x, int64_t xs, int64_t n, int width,
                                                           float* __restrict__ inv, float* __restrict__ out, int64_t os) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, l = lane % LPR, col = 4 * l;
    const bool active = col < width;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ngroups = (n + RPW - 1) / RPW;
    for (int64_t g = wave0; g < ngroups; g += nwaves) {
        const int64_t j = g * RPW + sub;
        const int64_t jc = j < n ? j : n - 1;
        const f32x4 m = mask_cols(*(const f32x4*)(x + jc * xs + (active ? col : 0)), active ? col : width, width);
        float ss = m[0] * m[0];
        ss = fmaf(m[1], m[1], ss); ss = fmaf(m[2], m[2], ss); ss = fmaf(m[3], m[3], ss);
        ss = group_sum<LPR>(ss);
        const float iv = 1.f / fmaxf(sqrtf(ss), MKGNN_EPS);
        if (j < n && l == 0) inv[j] = iv;
        if (j < n && active) *(f32x4*)(out + j * os + col) = split_row_store(m, iv);
    }
}

static inline int csr_grid(int64_t n, int rpw) {
    int64_t groups = (n + rpw - 1) / rpw;
    int64_t blocks = (groups + 3) / 4;
    if (blocks < 1) blocks = 1;
    static const char* env_cap = getenv("MKGNN_CSR_BLOCKS");          // diagnostics
    const int64_t cap = env_cap ? atoi(env_cap) : 256 * 16;
    if (blocks > cap) blocks = cap;
    return (int)blocks;
}

static inline int lanes_per_row(int width) { return width <= 32 ? 8 : width <= 64 ? 16 : width <= 128 ? 32 : 64; }

bool aligned_rows(const void* p, int64_t stride, int width) {
    return ((uintptr_t)p & 15) == 0 && stride % 4 == 0 && stride >= (width + 3) / 4 * 4;
}

bool try_rows_presplit(const float* x, int64_t xs, int64_t n, int width, float* inv, float* out, int64_t os, hipStream_t st, hipError_t* err) {
    if (n == 0 || width > 256 || !aligned_rows(x, xs, width) || !aligned_rows(out, os, width)) return false;
    const int lpr = lanes_per_row(width);
    const int64_t groups = (n + 64 / lpr - 1) / (64 / lpr);
    int64_t blocks = (groups + 3) / 4;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (lpr) {
        case 8: row_presplit_kernel<8><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv, out, os); break;
        case 16: row_presplit_kernel<16><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv, out, os); break;
        case 32: row_presplit_kernel<32><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv, out, os); break;
        default: row_presplit_kernel<64><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv, out, os); break;
    }
    *err = hipGetLastError();
    return true;
}

static int csr_fixed4() {
    static const int v = [] { const char* e = getenv("MKGNN_CSR_FIXED4"); return e ? atoi(e) : 0; }();
    return v;
}

template <bool GATHER>
static hipError_t launch_csr(const CsrArgs& a_in, hipStream_t st) {
    CsrArgs a = a_in;
    a.fixed4 = csr_fixed4();
    switch (lanes_per_row(a.width)) {
        case 8: csr_rows_kernel<8, GATHER><<<csr_grid(a.n, 8), 256, 0, st>>>(a); break;
        case 16: csr_rows_kernel<16, GATHER><<<csr_grid(a.n, 4), 256, 0, st>>>(a); break;
        case 32: csr_rows_kernel<32, GATHER><<<csr_grid(a.n, 2), 256, 0, st>>>(a); break;
        default: csr_rows_kernel<64, GATHER><<<csr_grid(a.n, 1), 256, 0, st>>>(a); break;
    }
    return hipGetLastError();
}

template <int BLK>
static hipError_t launch_csr_blocks(const CsrArgs& a_in, hipStream_t st) {
    CsrArgs a = a_in;
    a.fixed4 = csr_fixed4();
    switch (lanes_per_row(a.width)) {
        case 8: csr_rows_kernel<8, false, BLK><<<csr_grid(a.n, 8), 256, 0, st>>>(a); break;
        case 16: csr_rows_kernel<16, false, BLK><<<csr_grid(a.n, 4), 256, 0, st>>>(a); break;
        case 32: csr_rows_kernel<32, false, BLK><<<csr_grid(a.n, 2), 256, 0, st>>>(a); break;
        default: csr_rows_kernel<64, false, BLK><<<csr_grid(a.n, 1), 256, 0, st>>>(a); break;
    }
    return hipGetLastError();
}

// Block-row segment sums (see csr_rows_kernel): no fallback, the caller checks segment_sum_blocks_supported first.
bool segment_sum_blocks_supported(const float* in, int64_t is, int64_t n, int width, const float* out, int64_t os) {
    return n > 0 && n < (1 << 28) && width <= 255 && aligned_rows(in, is, width) && aligned_rows(out, os, width);
}

hipError_t launch_segment_sum_blocks(const float* in, int64_t is, const int32_t* rowptr, const int32_t* col, const int8_t* deg8,
                                     int64_t n, int width, const int32_t L[4], int mode, float* out, int64_t os,
                                     float* inv_norm, hipStream_t st) {
    CsrArgs a{};
    a.src = in; a.ss = is; a.rowptr = rowptr; a.idx = col; a.n = n; a.width = width; a.out = out; a.os = os;
    a.inv_out = (mode == 1 || mode == 3) ? inv_norm : nullptr;
    a.split_out = mode == 3 ? 1 : 0;
    a.deg8 = deg8;
    int off = 0;
    for (int d = 1; d <= 4; ++d) {
        a.blk_off |= (uint64_t)off << (8 * d);
        a.blk_len |= (uint64_t)L[d - 1] << (8 * d);
        off += L[d - 1];
    }
    return mode != 2 ? launch_csr_blocks<1>(a, st) : launch_csr_blocks<2>(a, st);
}

// Fast paths; the callers fall back to the one-row-per-wave kernels when these decline (return false).
bool try_segment_sum_aligned(const float* in, int64_t is, const int32_t* rowptr, const int32_t* col, int64_t n, int width,
                             float* out, int64_t os, float* inv_norm, hipStream_t st, hipError_t* err) {
    if (n == 0 || n >= (1 << 30) || !col || width > 256 || !aligned_rows(in, is, width) || !aligned_rows(out, os, width)) return false;
    CsrArgs a{};
    a.src = in; a.ss = is; a.rowptr = rowptr; a.idx = col; a.n = n; a.width = width; a.out = out; a.os = os;
    a.inv_out = inv_norm;
    *err = launch_csr<false>(a, st);
    return true;
}

bool try_backward_gather_aligned(const float* contrib, int64_t cs, const int32_t* rowptr, const int32_t* rows, const float* x,
                                 int64_t xs, const float* inv, int64_t n, int F, float* gx, int64_t gxs, hipStream_t st,
                                 hipError_t* err, bool x_split) {
    if (n == 0 || n >= (1 << 30) || !rows || F > 256 || !aligned_rows(contrib, cs, F) || !aligned_rows(x, xs, F) || !aligned_rows(gx, gxs, F))
        return false;
    CsrArgs a{};
    a.src = contrib; a.ss = cs; a.rowptr = rowptr; a.idx = rows; a.n = n; a.width = F; a.out = gx; a.os = gxs;
    a.x = x; a.xs = xs; a.inv = inv; a.x_split = x_split ? 1 : 0;
    *err = launch_csr<true>(a, st);
    return true;
}

bool try_row_inv_norm_aligned(const float* x, int64_t xs, int64_t n, int width, float* inv, hipStream_t st, hipError_t* err) {
    if (n == 0 || width > 256 || !aligned_rows(x, xs, width)) return false;
    const int lpr = lanes_per_row(width);
    const int64_t groups = (n + 64 / lpr - 1) / (64 / lpr);
    int64_t blocks = (groups + 7) / 8;      // two groups per wave iteration
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (lpr) {
        case 8: row_inv_norm_aligned_kernel<8><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv); break;
        case 16: row_inv_norm_aligned_kernel<16><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv); break;
        case 32: row_inv_norm_aligned_kernel<32><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv); break;
        default: row_inv_norm_aligned_kernel<64><<<(int)blocks, 256, 0, st>>>(x, xs, n, width, inv); break;
    }
    *err = hipGetLastError();
    return true;
}

}  // namespace mkgnn
