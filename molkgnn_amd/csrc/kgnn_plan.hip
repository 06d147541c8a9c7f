// Per-batch index plan on the GPU (molkgnn_amd/plan.py): the scatter CSR of the backward pass and the two CSR forms of
// edge_index for MolGCN.propagate, built from the degree buckets and the edge list without a sort over the whole
// batch and without a host round trip (so a batch can be planned inside a captured graph or a timed step).
//
// The torch builder (plan.py: three stable sorts + bincount + cumsum) defines the result: inside every atom's segment
// the entries are in ascending original order (contribution-row number / edge number).  Here: count per atom with
// integer atomics; exclusive scan; every entry takes a slot of its atom's segment with an integer atomic (the slot
// ORDER is arbitrary, the segment's CONTENT is not); one thread per atom then sorts its few entries -- deterministic
// whatever the atomics did, and equal to the torch builder entry for entry.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "kgnn_common.h"
#include "kgnn_launch.h"

namespace mkgnn {

struct PlanArgs {
    const int64_t* sel[4]; const int64_t* nei[4];
    int64_t count[4];
    int64_t row_base[5];         // contribution rows of bucket d start at row_base[d]
    const int64_t* edge_index;   // [2, M]
    int64_t n, m, r;
    int32_t* cnt;                // [3][n + 1] counts, then cursors
    int32_t* tmp[3];             // unsorted segment contents: [r], [m], [m]
    int32_t* rowptr[3];          // outputs: scatter, in (by target), out (by source)
    int32_t* scatter_rows; int32_t* in_col; int32_t* in_col_packed; int32_t* out_col;
    int8_t* deg8;
};

// destination atom of contribution row i (rows: bucket by bucket, atom by atom, focal then neighbours)
__device__ __forceinline__ int64_t plan_row_dest(const PlanArgs& a, int64_t i) {
    int d = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (i >= a.row_base[k]) d = k;
    const int64_t j = i - a.row_base[d];
    const int64_t nloc = j / (d + 2), s = j - nloc * (d + 2);
    return s == 0 ? a.sel[d][nloc] : a.nei[d][nloc * (d + 1) + s - 1];
}

template <bool FILL>
__global__ void __launch_bounds__(256) plan_count_fill_kernel(PlanArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < a.r) {
        const int64_t dest = plan_row_dest(a, i);
        if (dest >= 0 && dest < a.n) {
            const int pos = atomicAdd(&a.cnt[dest], 1);
            if (FILL) a.tmp[0][pos] = (int32_t)i;
        }
    }
    if (i < a.m) {
        const int64_t s = a.edge_index[i], t = a.edge_index[a.m + i];
        if (s >= 0 && s < a.n && t >= 0 && t < a.n) {
            const int p1 = atomicAdd(&a.cnt[(a.n + 1) + t], 1);
            const int p2 = atomicAdd(&a.cnt[2 * (a.n + 1) + s], 1);
            if (FILL) { a.tmp[1][p1] = (int32_t)i; a.tmp[2][p2] = (int32_t)i; }
        }
    }
    // the degree bucket of every atom (0 = none): zeroed by the counting pass, scattered by the fill pass (no memset launch)
    if (!FILL) {
        if (i < a.n) a.deg8[i] = 0;
    } else {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (i < a.count[d]) a.deg8[a.sel[d][i]] = (int8_t)(d + 1);
    }
}

// exclusive scan of cnt[k][0..n) -> rowptr[k][0..n] and, in place, the fill pass's cursors.  Two coalesced passes:
// totals of 2048-element blocks, then every block scans its elements behind the sum of the totals before it.
constexpr int PLAN_SCAN_ELEMS = 2048;                    // per block: 256 threads x 8 chunks of 256 consecutive elements

__device__ __forceinline__ int plan_block_scan256(int v, int* sh, int& total) {     // exclusive scan over the block's 256 threads
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

__global__ void __launch_bounds__(256) plan_blocksum_kernel(PlanArgs a, int32_t* bsum, int nblk) {
    __shared__ int sh[4];
    const int k = blockIdx.y;
    const int32_t* c = a.cnt + (size_t)k * (a.n + 1);
    const int64_t base = (int64_t)blockIdx.x * PLAN_SCAN_ELEMS;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + 256 * j + threadIdx.x;
        s += i < a.n ? c[i] : 0;
    }
    int total;
    (void)plan_block_scan256(s, sh, total);
    if (threadIdx.x == 0) bsum[k * nblk + blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) plan_scan_kernel(PlanArgs a, const int32_t* bsum, int nblk) {
    __shared__ int sh[4];
    const int k = blockIdx.y;
    int32_t* c = a.cnt + (size_t)k * (a.n + 1);
    // sum of the block totals before this block (a few dozen values; every thread reads them all, broadcast loads)
    int carry = 0;
    for (int b = 0; b < (int)blockIdx.x; ++b) carry += bsum[k * nblk + b];
    const int64_t base = (int64_t)blockIdx.x * PLAN_SCAN_ELEMS;
    for (int j = 0; j < 8; ++j) {
        const int64_t i = base + 256 * j + threadIdx.x;
        const int v = i < a.n ? c[i] : 0;
        int total;
        const int ex = plan_block_scan256(v, sh, total);
        if (i < a.n) { c[i] = carry + ex; a.rowptr[k][i] = carry + ex; }
        carry += total;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.rowptr[k][a.n] = carry;
}

// one thread per (array, atom): sort the segment, then the stored form of every entry
__global__ void __launch_bounds__(256) plan_sort_kernel(PlanArgs a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= 3 * a.n) return;
    const int k = (int)(g / a.n);
    const int64_t atom = g - (int64_t)k * a.n;
    const int lo = a.rowptr[k][atom], hi = a.rowptr[k][atom + 1];
    int32_t* t = a.tmp[k];
    for (int i = lo + 1; i < hi; ++i) {                 // insertion sort (segments hold a handful of entries)
        const int32_t v = t[i];
        int j = i - 1;
        while (j >= lo && t[j] > v) { t[j + 1] = t[j]; --j; }
        t[j + 1] = v;
    }
    for (int i = lo; i < hi; ++i) {
        const int32_t e = t[i];
        if (k == 0) a.scatter_rows[i] = e;
        else if (k == 1) {
            const int32_t src = (int32_t)a.edge_index[e];
            a.in_col[i] = src;
            if (a.in_col_packed) a.in_col_packed[i] = src | ((int32_t)a.deg8[src] << 28);
        } else a.out_col[i] = (int32_t)a.edge_index[a.m + e];
    }
}

// compact wire form -> collated batch tensors (mkgnn_expand_batch): thread t < n_bonds expands bond t, thread t < n_atoms
// finds atom t's molecule by bisection of mol_ptr
__global__ void __launch_bounds__(256) expand_batch_kernel(const int32_t* __restrict__ bond_ij, const uint8_t* __restrict__ bond_attr,
                                                           int64_t n_bonds, int E, const int32_t* __restrict__ mol_ptr,
                                                           int64_t n_mol, int64_t n_atoms, int64_t* __restrict__ edge_index,
                                                           float* __restrict__ edge_attr, int64_t* __restrict__ batch,
                                                           int32_t* __restrict__ atom_mol) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_bonds) {
        const int64_t i = bond_ij[2 * t], j = bond_ij[2 * t + 1], m = 2 * n_bonds;
        if (edge_index) {
            edge_index[2 * t] = i; edge_index[2 * t + 1] = j;
            edge_index[m + 2 * t] = j; edge_index[m + 2 * t + 1] = i;
        }
        if (edge_attr && bond_attr)
            for (int k = 0; k < E; ++k) {
                const float v = (float)bond_attr[t * E + k];
                edge_attr[(2 * t) * E + k] = v;
                edge_attr[(2 * t + 1) * E + k] = v;
            }
    }
    if (t < n_atoms && mol_ptr && (batch || atom_mol)) {
        int64_t lo = 0, hi = n_mol;                  // the molecule g with mol_ptr[g] <= t < mol_ptr[g + 1]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)mol_ptr[mid] <= t) lo = mid; else hi = mid;
        }
        if (batch) batch[t] = lo;
        if (atom_mol) atom_mol[t] = (int32_t)lo;
    }
}

}  // namespace mkgnn

extern "C" int mkgnn_expand_batch(const int32_t* bond_ij, const uint8_t* bond_attr, int64_t n_bonds, int32_t E,
                                  const int32_t* mol_ptr, int64_t n_molecules, int64_t n_atoms, int64_t* edge_index,
                                  float* edge_attr, int64_t* batch, int32_t* atom_molecule, void* stream) {
    using namespace mkgnn;
    if (n_bonds < 0 || n_atoms < 0 || n_molecules < 0 || E < 0) return api_fail("mkgnn_expand_batch: negative size");
    if (n_bonds && !bond_ij) return api_fail("mkgnn_expand_batch: bond_ij is null");
    if ((batch || atom_molecule) && n_atoms && (!mol_ptr || n_molecules < 1)) return api_fail("mkgnn_expand_batch: mol_ptr is needed for batch / atom_molecule");
    const int64_t n = n_bonds > n_atoms ? n_bonds : n_atoms;
    if (n == 0) return 0;
    expand_batch_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(bond_ij, bond_attr, n_bonds, E, mol_ptr, n_molecules,
                                                                                     n_atoms, edge_index, edge_attr, batch, atom_molecule);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_fail("mkgnn_expand_batch: launch failed");
}


using namespace mkgnn;

extern "C" size_t mkgnn_plan_workspace_bytes(int64_t n_atoms, int64_t n_edges, int64_t n_rows) {
    return (size_t)(3 * (n_atoms + 1) + n_rows + 2 * n_edges + 3 * (n_atoms / PLAN_SCAN_ELEMS + 2)) * 4 + 1024;
}

extern "C" int mkgnn_plan_build(const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], int64_t n_atoms, const int64_t* edge_index,
                                int64_t n_edges, int32_t* scatter_rowptr, int32_t* scatter_rows, int32_t* in_rowptr, int32_t* in_col,
                                int32_t* in_col_packed, int32_t* out_rowptr, int32_t* out_col, int8_t* deg8, void* workspace,
                                size_t workspace_bytes, void* stream) {
    const char* who = "mkgnn_plan_build";
    if (!buckets || n_atoms < 0 || n_edges < 0) return api_fail("%s: bad arguments", who);
    if (n_atoms >= (1ll << 28) || n_edges >= (1ll << 31)) return api_fail("%s: %lld atoms / %lld edges exceed the 32-bit plan", who, (long long)n_atoms, (long long)n_edges);
    PlanArgs a;
    int64_t r = 0;
    for (int d = 0; d < 4; ++d) {
        a.sel[d] = buckets[d].selected_index; a.nei[d] = buckets[d].nei_index; a.count[d] = buckets[d].count;
        if (a.count[d] < 0 || (a.count[d] > 0 && (!a.sel[d] || !a.nei[d]))) return api_fail("%s: degree %d bucket", who, d + 1);
        a.row_base[d] = r;
        r += a.count[d] * (d + 2);
    }
    a.row_base[4] = r;
    if (n_edges > 0 && !edge_index) return api_fail("%s: edge_index is null", who);
    if (!scatter_rowptr || !in_rowptr || !out_rowptr || !deg8 || (r > 0 && !scatter_rows) || (n_edges > 0 && (!in_col || !out_col)))
        return api_fail("%s: null output", who);
    if (!workspace || workspace_bytes < mkgnn_plan_workspace_bytes(n_atoms, n_edges, r))
        return api_fail("%s: workspace of %zu bytes, need %zu", who, workspace_bytes, mkgnn_plan_workspace_bytes(n_atoms, n_edges, r));
    a.edge_index = edge_index; a.n = n_atoms; a.m = n_edges; a.r = r;
    int32_t* w = (int32_t*)workspace;
    a.cnt = w; w += 3 * (n_atoms + 1);
    const int nblk_scan = (int)((n_atoms + PLAN_SCAN_ELEMS - 1) / PLAN_SCAN_ELEMS);
    int32_t* bsum = w; w += 3 * (nblk_scan > 0 ? nblk_scan : 1);
    a.tmp[0] = w; w += r; a.tmp[1] = w; w += n_edges; a.tmp[2] = w;
    a.rowptr[0] = scatter_rowptr; a.rowptr[1] = in_rowptr; a.rowptr[2] = out_rowptr;
    a.scatter_rows = scatter_rows; a.in_col = in_col; a.in_col_packed = in_col_packed; a.out_col = out_col; a.deg8 = deg8;
    hipStream_t st = (hipStream_t)stream;
    // (rounded up to 16 bytes: the tail lands in the block sums, which the scan's first pass writes before anyone reads them;
    // an odd size makes the runtime launch a second fill kernel)
    hipError_t e = hipMemsetAsync(a.cnt, 0, ((size_t)3 * (n_atoms + 1) * 4 + 15) & ~(size_t)15, st);
    if (e != hipSuccess) return api_hip_fail(who, e);
    int64_t work = r > n_edges ? r : n_edges;
    if (n_atoms > work) work = n_atoms;
    for (int d = 0; d < 4; ++d) if (a.count[d] > work) work = a.count[d];
    if (n_atoms == 0) return 0;
    const unsigned grid = (unsigned)((work + 255) / 256);
    if (grid) plan_count_fill_kernel<false><<<grid, 256, 0, st>>>(a);
    plan_blocksum_kernel<<<dim3(nblk_scan, 3), 256, 0, st>>>(a, bsum, nblk_scan);
    plan_scan_kernel<<<dim3(nblk_scan, 3), 256, 0, st>>>(a, bsum, nblk_scan);
    if (grid) plan_count_fill_kernel<true><<<grid, 256, 0, st>>>(a);
    plan_sort_kernel<<<(unsigned)((3 * n_atoms + 255) / 256), 256, 0, st>>>(a);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}
