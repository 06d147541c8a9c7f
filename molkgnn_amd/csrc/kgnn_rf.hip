// Receptive-field builder: the degree buckets the kernel convolution consumes, built on the GPU
// from a collated batch's edge list.  Restates, for a whole batch at once, what the reference computes per
// molecule in ToXAndPAndEdgeAttrForDeg (wrapper.py:559-672) followed by PyG collation:
//
//   selected_index_degD [N_d]        atoms whose out-degree is D, ascending atom id      (wrapper.py:574-576, 600-601)
//   nei_index_degD      [N_d * D]    targets of the atom's edges IN EDGE-LIST ORDER      (wrapper.py:567-572, 623-624)
//   nei_edge_attr_degD  [N_d * D, E] attributes of bond 2*(e/2) of each such edge e      (wrapper.py:578-593)
//   p_focal_degD [N_d, 3], nei_p_degD [N_d * D, 3]
//
// Two passes.  count: one thread per edge takes a slot of its source atom with an integer atomic (the slot ORDER is
// arbitrary, the slot CONTENT -- up to four edge ids -- is not), then per-block bucket sizes and their scan give every
// atom its rank inside its bucket.  fill: one thread per atom sorts its <= 4 edge ids (restoring edge-list order:
// deterministic whatever the atomics did) and writes its rows.  No sort over the edge list, no host round trip
// except the four bucket sizes the caller needs to allocate the outputs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "kgnn_common.h"
#include "kgnn_launch.h"
#include "../../include/molkgnn_hip.h"

namespace mkgnn {

constexpr int RF_BLOCK = 256;

struct RfWs {
    int32_t* deg;        // [n]      out-degree
    int32_t* slot;       // [n * 4]  first four edge ids of every atom (any order)
    int32_t* blk;        // [nblk * 4] per-block bucket sizes, then their exclusive scan
    int64_t* total;      // [4]
};

static size_t rf_ws_bytes(int64_t n) {
    const int64_t nblk = (n + RF_BLOCK - 1) / RF_BLOCK;
    return (size_t)(n * 4 + n * 16 + nblk * 16 + 64 + 256);
}

static RfWs rf_ws(void* ws, int64_t n) {
    const int64_t nblk = (n + RF_BLOCK - 1) / RF_BLOCK;
    char* p = (char*)ws;
    RfWs w;
    w.total = (int64_t*)p; p += 64;
    w.deg = (int32_t*)p; p += n * 4;
    p = (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    w.slot = (int32_t*)p; p += n * 16;
    w.blk = (int32_t*)p; p += nblk * 16;
    return w;
}

__global__ void __launch_bounds__(256) rf_edges_kernel(const int64_t* __restrict__ src, int64_t n, int64_t m, RfWs w) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m) return;
    const int64_t s = src[e];
    if (s < 0 || s >= n) return;
    const int c = atomicAdd(&w.deg[s], 1);
    if (c < 4) w.slot[s * 4 + c] = (int32_t)e;
}

// bucket of atom i: 0..3 for out-degree 1..4, -1 otherwise
__device__ __forceinline__ int rf_bucket(int deg) { return (deg >= 1 && deg <= 4) ? deg - 1 : -1; }

__global__ void __launch_bounds__(RF_BLOCK) rf_block_count_kernel(int64_t n, RfWs w) {
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * RF_BLOCK + threadIdx.x;
    const int b = i < n ? rf_bucket(w.deg[i]) : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long mask = __ballot(b == k);
        if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&cnt[k], __popcll(mask));      // integer: order-independent
    }
    __syncthreads();
    if (threadIdx.x < 4) w.blk[(int64_t)blockIdx.x * 4 + threadIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of the per-block sizes (one block; nblk is a few hundred to a few thousand)
__global__ void __launch_bounds__(256) rf_scan_kernel(int64_t nblk, RfWs w, int64_t* counts) {
    __shared__ int part[4][256];
    const int t = threadIdx.x;
    const int64_t per = (nblk + 255) / 256;
    const int64_t lo = per * t, hi = lo + per < nblk ? lo + per : nblk;
    int s[4] = {0, 0, 0, 0};
    for (int64_t b = lo; b < hi; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += w.blk[b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) part[k][t] = s[k];
    __syncthreads();
    if (t < 4) {                                     // serial scan over 256 partials per bucket: trivial
        int run = 0;
        for (int j = 0; j < 256; ++j) { const int v = part[t][j]; part[t][j] = run; run += v; }
        w.total[t] = run;
        if (counts) counts[t] = run;
    }
    __syncthreads();
    int run[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) run[k] = part[k][t];
    for (int64_t b = lo; b < hi; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int v = w.blk[b * 4 + k]; w.blk[b * 4 + k] = run[k]; run[k] += v; }
}

struct RfOut {
    int64_t* sel[4]; int64_t* nei[4]; float* eattr[4]; float* pf[4]; float* pn[4];
    float* eunit[4];     // optional: [rows, 8] unit-normalised bond rows (mkgnn_degree_bucket.nei_edge_unit), written with the raw ones
    int64_t cap[4];      // rows the caller allocated per bucket (ranks beyond it are dropped, never written)
};

__global__ void __launch_bounds__(RF_BLOCK) rf_fill_kernel(const int64_t* __restrict__ dst, const float* __restrict__ p,
                                                           const float* __restrict__ eattr, int64_t n, int E, RfWs w, RfOut o) {
    __shared__ int wave_cnt[4][RF_BLOCK / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * RF_BLOCK + t;
    const int deg = i < n ? w.deg[i] : 0;
    const int b = i < n ? rf_bucket(deg) : -1;
    int below = 0;                                   // atoms of my bucket before me in this wave
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long mask = __ballot(b == k);
        if (b == k) below = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[k][wave] = __popcll(mask);
    }
    __syncthreads();
    // rows the caller allocated beyond the batch's real bucket sizes (a fixed-shape caller whose `sizes` are larger than
    // the degree counts): zero-filled -- atom 0, zero attributes -- so that nothing downstream gathers through
    // uninitialised indices; mkgnn_rf_count's `counts` says what the real sizes were (receptive_field.check_sizes)
    {
        const int64_t nthreads = (int64_t)gridDim.x * RF_BLOCK;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dk = k + 1;
            for (int64_t rr = w.total[k] + i; rr < o.cap[k]; rr += nthreads) {
                o.sel[k][rr] = 0;
                for (int c = 0; c < 3; ++c) o.pf[k][rr * 3 + c] = 0.f;
                for (int s_ = 0; s_ < dk; ++s_) {
                    const int64_t row = rr * dk + s_;
                    o.nei[k][row] = 0;
                    for (int c = 0; c < 3; ++c) o.pn[k][row * 3 + c] = 0.f;
                    for (int c = 0; c < E; ++c) o.eattr[k][row * E + c] = 0.f;
                    if (o.eunit[k] && E <= 8) for (int c = 0; c < 8; ++c) o.eunit[k][row * 8 + c] = 0.f;
                }
            }
        }
    }
    if (b < 0) return;
    int r = w.blk[(int64_t)blockIdx.x * 4 + b] + below;
    for (int k = 0; k < wave; ++k) r += wave_cnt[b][k];
    if (r >= o.cap[b]) return;
    // my edges, back in edge-list order
    int e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = k < deg ? w.slot[i * 4 + k] : 0x7fffffff;
#define RF_CSWAP(a, c) { const int lo_ = e[a] < e[c] ? e[a] : e[c], hi_ = e[a] < e[c] ? e[c] : e[a]; e[a] = lo_; e[c] = hi_; }
    RF_CSWAP(0, 1) RF_CSWAP(2, 3) RF_CSWAP(0, 2) RF_CSWAP(1, 3) RF_CSWAP(1, 2)
#undef RF_CSWAP
    const int d = deg;
    o.sel[b][r] = i;
    float* pf = o.pf[b] + (int64_t)r * 3;
    pf[0] = p[i * 3]; pf[1] = p[i * 3 + 1]; pf[2] = p[i * 3 + 2];
    for (int k = 0; k < d; ++k) {
        const int64_t row = (int64_t)r * d + k;
        const int64_t j = dst[e[k]];
        o.nei[b][row] = j;
        const int64_t jc = j < 0 ? 0 : (j >= n ? n - 1 : j);           // a bad target stays visible in nei_index, reads stay in bounds
        float* pn = o.pn[b] + row * 3;
        pn[0] = p[jc * 3]; pn[1] = p[jc * 3 + 1]; pn[2] = p[jc * 3 + 2];
        const float* src = eattr + (int64_t)(e[k] & ~1) * E;            // both directions of a bond share edge 2*(e/2)
        float* ea = o.eattr[b] + row * E;
        for (int c = 0; c < E; ++c) ea[c] = src[c];
        if (o.eunit[b] && E <= 8) {
            // the arithmetic of mkgnn_unit_rows8 (kgnn_fwd_stream.hip unit_rows8_kernel), so that the two agree bit for bit
            float ev[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) ev[c] = c < E ? src[c] : 0.f;
            float pq[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) pq[c] = fmaf(ev[2 * c + 1], ev[2 * c + 1], ev[2 * c] * ev[2 * c]);
            const float s2 = __fadd_rn(__fadd_rn(pq[0], pq[1]), __fadd_rn(pq[2], pq[3]));
            const float ie = 1.f / fmaxf(sqrtf(s2), MKGNN_EPS);
            float* eu = o.eunit[b] + row * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) eu[c] = ev[c] * ie;
        }
    }
}

}  // namespace mkgnn

using namespace mkgnn;

extern "C" {

size_t mkgnn_rf_workspace_bytes(int64_t n_atoms) { return n_atoms > 0 ? rf_ws_bytes(n_atoms) : 256; }

int mkgnn_rf_count(const int64_t* edge_index, int64_t n_atoms, int64_t n_edges, void* workspace, size_t workspace_bytes,
                   int64_t* counts, void* stream) {
    if (n_atoms < 0 || n_edges < 0 || n_atoms >= ((int64_t)1 << 31) || n_edges >= ((int64_t)1 << 31))
        return api_fail("mkgnn_rf_count: sizes outside 0..2^31");
    if (!counts || !workspace || workspace_bytes < mkgnn_rf_workspace_bytes(n_atoms) || ((uintptr_t)workspace & 15))
        return api_fail("mkgnn_rf_count: null / small / misaligned workspace or counts");
    if (n_edges && !edge_index) return api_fail("mkgnn_rf_count: edge_index is null");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (n_atoms == 0) {
        e = hipMemsetAsync(counts, 0, 4 * sizeof(int64_t), st);
        return e == hipSuccess ? 0 : api_hip_fail("mkgnn_rf_count", e);
    }
    RfWs w = rf_ws(workspace, n_atoms);
    // (rounded up to 16 bytes -- the alignment padding in front of `slot` absorbs it: an odd size makes the runtime launch a
    // second fill kernel for the tail)
    e = hipMemsetAsync(w.deg, 0, ((size_t)n_atoms * 4 + 15) & ~(size_t)15, st);
    if (e != hipSuccess) return api_hip_fail("mkgnn_rf_count", e);
    const int64_t nblk = (n_atoms + RF_BLOCK - 1) / RF_BLOCK;
    if (n_edges) rf_edges_kernel<<<(int)((n_edges + 255) / 256), 256, 0, st>>>(edge_index, n_atoms, n_edges, w);
    rf_block_count_kernel<<<(int)nblk, RF_BLOCK, 0, st>>>(n_atoms, w);
    rf_scan_kernel<<<1, 256, 0, st>>>(nblk, w, counts);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_rf_count", e);
}

int mkgnn_rf_fill(const int64_t* edge_index, const float* p, const float* edge_attr, int64_t n_atoms, int64_t n_edges,
                  int32_t E, const void* workspace, const mkgnn_degree_bucket out[MKGNN_MAX_DEGREE], void* stream) {
    if (n_atoms <= 0) return 0;
    if (!workspace || !out || !p || E < 1) return api_fail("mkgnn_rf_fill: null pointer or bad E");
    if (n_edges && (!edge_index || !edge_attr)) return api_fail("mkgnn_rf_fill: edge arrays are null");
    RfWs w = rf_ws((void*)workspace, n_atoms);
    RfOut o;
    for (int k = 0; k < 4; ++k) {
        o.sel[k] = (int64_t*)out[k].selected_index; o.nei[k] = (int64_t*)out[k].nei_index;
        o.eattr[k] = (float*)out[k].nei_edge_attr; o.pf[k] = (float*)out[k].p_focal; o.pn[k] = (float*)out[k].nei_p;
        o.eunit[k] = (float*)out[k].nei_edge_unit;
        o.cap[k] = out[k].count;
        if (out[k].count > 0 && (!o.sel[k] || !o.nei[k] || !o.eattr[k] || !o.pf[k] || !o.pn[k]))
            return api_fail("mkgnn_rf_fill: degree %d has %lld atoms but null outputs", k + 1, (long long)out[k].count);
    }
    const int64_t nblk = (n_atoms + RF_BLOCK - 1) / RF_BLOCK;
    rf_fill_kernel<<<(int)nblk, RF_BLOCK, 0, (hipStream_t)stream>>>(edge_index + n_edges, p, edge_attr, n_atoms, E, w, o);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_rf_fill", e);
}

}  // extern "C"
