// Receptive fields AND index plan of a collated batch in one pass (round 5): what mkgnn_rf_count +
// mkgnn_rf_fill (kgnn_rf.hip: the reference's ToXAndPAndEdgeAttrForDeg, wrapper.py:559-672, for a whole batch) and
// mkgnn_plan_build (kgnn_plan.hip: the backward's scatter CSR, propagate's two CSRs, deg8) compute in 9 kernels and 2
// memsets -- counting the same degrees twice and scanning twice -- as ONE memset and 6 kernels:
//
//   edges     one thread per edge: out-degree of the source + a slot for the edge id (integer atomics: the slot ORDER is
//             arbitrary, its CONTENT is not); in- / out-degree over the edges whose both ends are atoms
//   count     a block per 256-atom chunk: the chunk's sums of {atoms of degree 1..4, in-degrees, out-degrees}
//   scan      a block per chunk: its bucket bases; in / out row pointers and cursors of its atoms
//   fill      atoms: the receptive-field rows (ids sorted back into edge-list order), deg8, their rank in the bucket;
//             edges: a place in the in / out segments                       [the receptive fields are complete here]
//   segments  atoms: their in / out segments sorted and stored; how many contribution rows point at them; chunk sums
//   scatter   a block per chunk: scatter row pointers; every atom pulls its entries (own focal row + one neighbour row per
//             in-edge from a bucketed atom: the source's rank and the edge's position among the source's ids), sorts, stores
//
// Every output is, entry for entry, what the separate builders give (tests/test_hip_parity.py): integer atomics only
// choose places, every segment is sorted afterwards.  The caller gives the bucket capacities (a batch padded to a fixed
// shape knows them: no host round trip, capturable); counts[0..3] receive the real sizes, counts[4] the number of atoms that
// did not fit their bucket's capacity (must be 0: receptive_field.check_sizes).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "kgnn_common.h"
#include "kgnn_launch.h"
#include "../../include/molkgnn_hip.h"

namespace mkgnn {


struct IxArgs {
    const int64_t* src; const int64_t* dst; const float* p; const float* eattr;
    int64_t n, m, r; int E;
    int32_t *deg, *cin, *cS, *cout, *slot, *rank, *csum, *chunk, *tmpI, *tmpO;
    int nchunk;
    int64_t* counts;
    int64_t* sel[4]; int64_t* nei[4]; float* ea[4]; float* pf[4]; float* pn[4]; float* eu[4];
    int64_t cap[4]; int64_t row_base[5];
    int32_t *rowptrS, *rowptrI, *rowptrO, *scatter_rows, *in_col, *in_col_packed, *out_col;
    int8_t* deg8;
};

__device__ __forceinline__ int ix_bucket(int deg) { return (deg >= 1 && deg <= 4) ? deg - 1 : -1; }

__global__ void __launch_bounds__(256) ix_edges_kernel(IxArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.m) return;
    const int64_t s = a.src[e], t = a.dst[e];
    if (s >= 0 && s < a.n) {
        const int c = atomicAdd(&a.deg[s], 1);
        if (c < 4) a.slot[s * 4 + c] = (int32_t)e;
        if (t >= 0 && t < a.n) {                          // (the plan's CSRs hold an edge only when both ends are atoms)
            atomicAdd(&a.cin[t], 1);
            atomicAdd(&a.cout[s], 1);
        }
    }
}

// sum of v over the block's 256 threads, in every thread (sh: 4 ints per quantity, the caller syncs between uses)
__device__ __forceinline__ int ix_block_sum(int v, int* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// exclusive scan over the block's 256 threads (plan_block_scan256 of kgnn_plan.hip)
__device__ __forceinline__ int ix_block_scan(int v, int* sh, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    __syncthreads();
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wave) base += sh[w];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + inc - v;
}

// one thread per atom, a block per 256-atom chunk: the chunk's sums of {atoms of degree 1..4, in-degree, out-degree}
__global__ void __launch_bounds__(256) ix_count_kernel(IxArgs a) {
    __shared__ int sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int v[6] = {0, 0, 0, 0, 0, 0};
    if (i < a.n) {
        const int b = ix_bucket(a.deg[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (b == k) ? 1 : 0;
        v[4] = a.cin[i];
        v[5] = a.cout[i];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int tot = ix_block_sum(v[q], sh);
        if (threadIdx.x == 0) a.csum[(int64_t)blockIdx.x * 8 + q] = tot;
        __syncthreads();
    }
}

// sum over the chunks before this block's of column q of a [nchunk][8] table, in every thread
__device__ __forceinline__ int ix_carry(const int32_t* tab, int q, int* sh) {
    int s = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) s += tab[(int64_t)c * 8 + q];
    return ix_block_sum(s, sh);
}

// a block per chunk: the chunk's bucket bases; in / out row pointers and cursors of its atoms
__global__ void __launch_bounds__(256) ix_scan_kernel(IxArgs a) {
    __shared__ int sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int carry[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { carry[q] = ix_carry(a.csum, q, sh); __syncthreads(); }
    if (threadIdx.x < 4) {
        int c4 = carry[0];
        c4 = threadIdx.x == 1 ? carry[1] : c4; c4 = threadIdx.x == 2 ? carry[2] : c4; c4 = threadIdx.x == 3 ? carry[3] : c4;
        a.chunk[(int64_t)blockIdx.x * 4 + threadIdx.x] = c4;
    }
    const bool last = blockIdx.x == gridDim.x - 1;
    int total;
    const int vi = i < a.n ? a.cin[i] : 0;
    const int exi = ix_block_scan(vi, sh, total);
    if (i < a.n) { a.cin[i] = carry[4] + exi; a.rowptrI[i] = carry[4] + exi; }
    if (last && threadIdx.x == 0) a.rowptrI[a.n] = carry[4] + total;
    __syncthreads();
    const int vo = i < a.n ? a.cout[i] : 0;
    const int exo = ix_block_scan(vo, sh, total);
    if (i < a.n) { a.cout[i] = carry[5] + exo; a.rowptrO[i] = carry[5] + exo; }
    if (last && threadIdx.x == 0) {
        a.rowptrO[a.n] = carry[5] + total;
        const int32_t* mine = a.csum + (int64_t)blockIdx.x * 8;
        a.counts[0] = carry[0] + mine[0]; a.counts[1] = carry[1] + mine[1];
        a.counts[2] = carry[2] + mine[2]; a.counts[3] = carry[3] + mine[3];
        a.counts[4] = 0;                                 // (atoms beyond a bucket's capacity: counted by the fill pass)
        a.counts[5] = 0;                                 // (reserved: written so that no entry of the six is left as allocated)
    }
}

__global__ void __launch_bounds__(256) ix_fill_kernel(IxArgs a, int fill_blocks) {
    __shared__ int wave_cnt[4][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * 256 + t;
    // ---- edges: a place in the in / out segments
    if (i < a.m) {
        const int64_t s = a.src[i], d_ = a.dst[i];
        if (s >= 0 && s < a.n && d_ >= 0 && d_ < a.n) {
            a.tmpI[atomicAdd(&a.cin[d_], 1)] = (int32_t)i;
            a.tmpO[atomicAdd(&a.cout[s], 1)] = (int32_t)i;
        }
    }
    // ---- atoms: receptive-field rows (the arithmetic and the order of rf_fill_kernel, kgnn_rf.hip)
    const int deg = i < a.n ? a.deg[i] : 0;
    const int b = i < a.n ? ix_bucket(deg) : -1;
    int below = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long mask = __ballot(b == k);
        if (b == k) below = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[k][wave] = __popcll(mask);
    }
    __syncthreads();
    {   // rows the caller allocated beyond the batch's real bucket sizes: zero-filled (atom 0, zero attributes)
        const int64_t nthreads = (int64_t)fill_blocks * 256;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dk = k + 1;
            for (int64_t rr = a.counts[k] + i; rr < a.cap[k]; rr += nthreads) {
                a.sel[k][rr] = 0;
                for (int c = 0; c < 3; ++c) a.pf[k][rr * 3 + c] = 0.f;
                for (int s_ = 0; s_ < dk; ++s_) {
                    const int64_t row = rr * dk + s_;
                    a.nei[k][row] = 0;
                    for (int c = 0; c < 3; ++c) a.pn[k][row * 3 + c] = 0.f;
                    for (int c = 0; c < a.E; ++c) a.ea[k][row * a.E + c] = 0.f;
                    if (a.eu[k] && a.E <= 8) for (int c = 0; c < 8; ++c) a.eu[k][row * 8 + c] = 0.f;
                }
            }
        }
    }
    if (i < a.n) { a.deg8[i] = (int8_t)(b + 1); a.rank[i] = -1; }
    if (b < 0 || (int64_t)blockIdx.x * 256 >= a.n) return;
    int r = a.chunk[(int64_t)blockIdx.x * 4 + b] + below;
    for (int k = 0; k < wave; ++k) r += wave_cnt[b][k];
    if (r >= a.cap[b]) { atomicAdd((unsigned long long*)&a.counts[4], 1ull); return; }
    a.rank[i] = r;
    int e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = k < deg ? a.slot[i * 4 + k] : 0x7fffffff;
#define IX_CSWAP(x, y) { const int lo_ = e[x] < e[y] ? e[x] : e[y], hi_ = e[x] < e[y] ? e[y] : e[x]; e[x] = lo_; e[y] = hi_; }
    IX_CSWAP(0, 1) IX_CSWAP(2, 3) IX_CSWAP(0, 2) IX_CSWAP(1, 3) IX_CSWAP(1, 2)
#undef IX_CSWAP
    const int d = deg;
    a.sel[b][r] = i;
    float* pf = a.pf[b] + (int64_t)r * 3;
    pf[0] = a.p[i * 3]; pf[1] = a.p[i * 3 + 1]; pf[2] = a.p[i * 3 + 2];
    for (int k = 0; k < d; ++k) {
        const int64_t row = (int64_t)r * d + k;
        const int64_t j = a.dst[e[k]];
        a.nei[b][row] = j;
        const int64_t jc = j < 0 ? 0 : (j >= a.n ? a.n - 1 : j);
        float* pn = a.pn[b] + row * 3;
        pn[0] = a.p[jc * 3]; pn[1] = a.p[jc * 3 + 1]; pn[2] = a.p[jc * 3 + 2];
        const float* src = a.eattr + (int64_t)(e[k] & ~1) * a.E;       // both directions of a bond share edge 2*(e/2)
        float* ea = a.ea[b] + row * a.E;
        for (int c = 0; c < a.E; ++c) ea[c] = src[c];
        if (a.eu[b] && a.E <= 8) {
            // the arithmetic of mkgnn_unit_rows8 (kgnn_fwd_stream.hip unit_rows8_kernel): bit for bit
            float ev[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) ev[c] = c < a.E ? src[c] : 0.f;
            float pq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) pq[c] = fmaf(ev[2 * c + 1], ev[2 * c + 1], ev[2 * c] * ev[2 * c]);
            const float s2 = __fadd_rn(__fadd_rn(pq[0], pq[1]), __fadd_rn(pq[2], pq[3]));
            const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
            float* eu = a.eu[b] + row * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) eu[c] = ev[c] * ie;
        }
    }
}

__device__ __forceinline__ void ix_sort_segment(int32_t* t, int lo, int hi) {      // (segments hold a handful of entries)
    for (int i = lo + 1; i < hi; ++i) {
        const int32_t v = t[i];
        int j = i - 1;
        while (j >= lo && t[j] > v) { t[j + 1] = t[j]; --j; }
        t[j + 1] = v;
    }
}

// one thread per atom: its in / out segments sorted (ascending edge number) and stored; the number of contribution rows that
// point at it -- its own focal row and one neighbour row per in-edge whose source sits in a bucket -- and the chunk's sum of them
__global__ void __launch_bounds__(256) ix_segments_kernel(IxArgs a) {
    __shared__ int sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int nS = 0;
    if (i < a.n) {
        {
            const int lo = a.rowptrI[i], hi = a.rowptrI[i + 1];
            ix_sort_segment(a.tmpI, lo, hi);
            for (int q = lo; q < hi; ++q) {
                const int32_t s = (int32_t)a.src[a.tmpI[q]];
                a.in_col[q] = s;
                if (a.in_col_packed) a.in_col_packed[q] = s | ((int32_t)a.deg8[s] << 28);
                nS += a.rank[s] >= 0 ? 1 : 0;
            }
        }
        {
            const int lo = a.rowptrO[i], hi = a.rowptrO[i + 1];
            ix_sort_segment(a.tmpO, lo, hi);
            for (int q = lo; q < hi; ++q) a.out_col[q] = (int32_t)a.dst[a.tmpO[q]];
        }
        nS += a.rank[i] >= 0 ? 1 : 0;
        a.cS[i] = nS;
    }
    const int tot = ix_block_sum(nS, sh);
    if (threadIdx.x == 0) a.csum[(int64_t)blockIdx.x * 8 + 6] = tot;
}

// a block per chunk: the scatter row pointers of its atoms, and every atom PULLS its entries (no atomics): the focal row of
// the atom itself, and for every in-edge e = (j -> i) from a bucketed j the row of j's slot that e fills -- its position
// among j's (at most four) edge ids -- then sorts them (ascending row number: the torch definition's stable order)
__global__ void __launch_bounds__(256) ix_scatter_kernel(IxArgs a) {
    __shared__ int sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int carry = ix_carry(a.csum, 6, sh);
    __syncthreads();
    int total;
    const int v = i < a.n ? a.cS[i] : 0;
    const int ex = ix_block_scan(v, sh, total);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.rowptrS[a.n] = carry + total;
    if (i >= a.n) return;
    const int lo = carry + ex;
    a.rowptrS[i] = lo;
    int w = lo;
    if (a.rank[i] >= 0) {
        const int b = (int)a.deg8[i] - 1;
        a.scatter_rows[w++] = (int32_t)(a.row_base[b] + (int64_t)a.rank[i] * (b + 2));
    }
    for (int q = a.rowptrI[i]; q < a.rowptrI[i + 1]; ++q) {
        const int32_t e = a.tmpI[q];
        const int64_t j = a.src[e];
        const int rj = a.rank[j];
        if (rj < 0) continue;
        const int bj = (int)a.deg8[j] - 1;
        int pos = 0;
        for (int k = 0; k <= bj; ++k) pos += a.slot[j * 4 + k] < e ? 1 : 0;
        a.scatter_rows[w++] = (int32_t)(a.row_base[bj] + (int64_t)rj * (bj + 2) + 1 + pos);
    }
    ix_sort_segment(a.scatter_rows, lo, w);
}

static size_t ix_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace mkgnn

using namespace mkgnn;

extern "C" size_t mkgnn_index_workspace_bytes(int64_t n_atoms, int64_t n_edges, int64_t n_rows) {
    if (n_atoms < 0 || n_edges < 0 || n_rows < 0) return 0;
    const int64_t nchunk = (n_atoms + 255) / 256 + 1;
    size_t b = 0;
    b += ix_align((size_t)(3 * n_atoms + 2) * 4);        // deg | cin | cout (zeroed together)
    b += ix_align((size_t)(n_atoms + 1) * 4);            // cS
    b += ix_align((size_t)n_atoms * 16);                 // slot
    b += ix_align((size_t)n_atoms * 4);                  // rank
    b += ix_align((size_t)nchunk * 8 * 4);               // csum
    b += ix_align((size_t)nchunk * 4 * 4);               // chunk
    b += 2 * ix_align((size_t)n_edges * 4);
    return b + 256;
}

extern "C" int mkgnn_index_build(const int64_t* edge_index, const float* p, const float* edge_attr, int64_t n_atoms,
                                 int64_t n_edges, int32_t E, const mkgnn_degree_bucket out[MKGNN_MAX_DEGREE],
                                 int32_t* scatter_rowptr, int32_t* scatter_rows, int32_t* in_rowptr, int32_t* in_col,
                                 int32_t* in_col_packed, int32_t* out_rowptr, int32_t* out_col, int8_t* deg8, int64_t* counts,
                                 void* workspace, size_t workspace_bytes, void* rf_ready_event, void* stream) {
    const char* who = "mkgnn_index_build";
    if (n_atoms < 1 || n_edges < 0 || n_atoms >= (1ll << 28) || n_edges >= (1ll << 31)) return api_fail("%s: sizes outside 1..2^28 atoms / 0..2^31 edges", who);
    if (!out || !p || E < 1 || !counts) return api_fail("%s: null pointer or bad E", who);
    if (n_edges && (!edge_index || !edge_attr)) return api_fail("%s: edge arrays are null", who);
    IxArgs a{};
    int64_t r = 0;
    for (int k = 0; k < 4; ++k) {
        a.sel[k] = (int64_t*)out[k].selected_index; a.nei[k] = (int64_t*)out[k].nei_index;
        a.ea[k] = (float*)out[k].nei_edge_attr; a.pf[k] = (float*)out[k].p_focal; a.pn[k] = (float*)out[k].nei_p;
        a.eu[k] = (float*)out[k].nei_edge_unit;
        a.cap[k] = out[k].count;
        if (out[k].count < 0 || (out[k].count > 0 && (!a.sel[k] || !a.nei[k] || !a.ea[k] || !a.pf[k] || !a.pn[k])))
            return api_fail("%s: degree %d has %lld rows but null outputs", who, k + 1, (long long)out[k].count);
        a.row_base[k] = r;
        r += a.cap[k] * (k + 2);
    }
    a.row_base[4] = r;
    if (r >= (1ll << 31)) return api_fail("%s: %lld contribution rows exceed the 32-bit plan", who, (long long)r);
    if (!scatter_rowptr || !in_rowptr || !out_rowptr || !deg8 || (r > 0 && !scatter_rows) || (n_edges > 0 && (!in_col || !out_col)))
        return api_fail("%s: null output", who);
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < mkgnn_index_workspace_bytes(n_atoms, n_edges, r))
        return api_fail("%s: workspace of %zu bytes, need %zu (16-byte aligned)", who, workspace_bytes, mkgnn_index_workspace_bytes(n_atoms, n_edges, r));
    a.src = edge_index; a.dst = edge_index ? edge_index + n_edges : nullptr; a.p = p; a.eattr = edge_attr;
    a.n = n_atoms; a.m = n_edges; a.r = r; a.E = E;
    a.nchunk = (int)((n_atoms + 255) / 256);
    char* w = (char*)workspace;
    auto take = [&](size_t bytes) { char* q = w; w += ix_align(bytes); return q; };
    a.deg = (int32_t*)take((size_t)(3 * n_atoms + 2) * 4);
    a.cin = a.deg + n_atoms; a.cout = a.cin + (n_atoms + 1);
    a.cS = (int32_t*)take((size_t)(n_atoms + 1) * 4);
    a.slot = (int32_t*)take((size_t)n_atoms * 16);
    a.rank = (int32_t*)take((size_t)n_atoms * 4);
    a.csum = (int32_t*)take((size_t)(a.nchunk + 1) * 8 * 4);
    a.chunk = (int32_t*)take((size_t)(a.nchunk + 1) * 4 * 4);
    a.tmpI = (int32_t*)take((size_t)n_edges * 4); a.tmpO = (int32_t*)take((size_t)n_edges * 4);
    a.counts = counts;
    a.rowptrS = scatter_rowptr; a.rowptrI = in_rowptr; a.rowptrO = out_rowptr;
    a.scatter_rows = scatter_rows; a.in_col = in_col; a.in_col_packed = in_col_packed; a.out_col = out_col; a.deg8 = deg8;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(a.deg, 0, ((size_t)(3 * n_atoms + 2) * 4 + 15) & ~(size_t)15, st);
    if (e != hipSuccess) return api_hip_fail(who, e);
    if (n_edges) ix_edges_kernel<<<(unsigned)((n_edges + 255) / 256), 256, 0, st>>>(a);
    ix_count_kernel<<<a.nchunk, 256, 0, st>>>(a);
    ix_scan_kernel<<<a.nchunk, 256, 0, st>>>(a);
    const int64_t work = n_atoms > n_edges ? n_atoms : n_edges;
    const int fill_blocks = (int)((work + 255) / 256);
    ix_fill_kernel<<<fill_blocks, 256, 0, st>>>(a, fill_blocks);
    // (the receptive fields are complete here: a caller that waits for them only records rf_ready_event now)
    if (rf_ready_event) {
        e = hipEventRecord((hipEvent_t)rf_ready_event, st);
        if (e != hipSuccess) return api_hip_fail(who, e);
    }
    ix_segments_kernel<<<a.nchunk, 256, 0, st>>>(a);
    ix_scatter_kernel<<<a.nchunk, 256, 0, st>>>(a);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}
