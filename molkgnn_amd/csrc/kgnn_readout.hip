// The two consumers either side of the kernel-convolution stack:
//
//   readout     pool_g( lin2( dropout( swish( lin1(h) ) ) ) )      reference MolKGNNNet.py:144-146
//   batch norm  BatchNorm1d over the atom rows                      reference MolKGNNNet.py:115
//
// Readout.  lin2 and the add-pool are both linear, so the molecule sum is taken first and lin2 is
// applied to one row per molecule: out_g = W2 (sum_{n in g} keep_n * swish(W1 h_n + b1)) + |g| b2.
// That removes the [N, H] x [H, G] product and its two gradients; what is left per atom is the
// [16 atoms x F] x [F x H] tile product, which runs on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32), fed straight from global memory with 16-byte loads: an MFMA's k index may
// be any permutation as long as both operands use the same one, so lane (row r, k-slot q) takes the
// four consecutive columns 16 j + 4 q .. + 3 of its row for the four k-steps of chunk j.
//
// Backward per 16-atom tile: dpre = dA[mol] * keep * swish'(pre) in registers, then two tile products,
// dh = dpre W1 (stored) and dW1 += dpre^T h (kept in accumulators for the whole kernel), followed by a
// fixed-order reduction block -> slab -> parameter.  No float atomics anywhere: results are reproducible.
//
// Batch norm.  Tall and skinny ([1e5, 28]): three short launches (column sums, centred squares, apply),
// every block re-deriving the column statistics from the per-block partials in a fixed order.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "kgnn_common.h"
#include <mutex>

#include "kgnn_launch.h"
#include "kgnn_philox.h"
#include "kgnn_prepare.h"
#include "kgnn_split.h"
#include "../../include/molkgnn_hip.h"

namespace mkgnn {

typedef mkgnn_f32x4 f32x4;

struct ReadoutArgs {
    const float* h; int64_t hs; int64_t n;
    const int32_t* mol_ptr; const int32_t* atom_mol; int64_t nmol;
    const float *w1, *b1, *w2, *b2;
    int F, H, G;
    const float* keep;             // [n, H] dropout multipliers or null
    float* pre;                    // [n, HP]
    float* pooled;                 // [nmol, HP]
    float* out; int64_t os;
    const float* gout; int64_t gos;
    float* dA;                     // [nmol, HP]
    float* gh; int64_t ghs;
    float* slab_atoms; int slab_atoms_stride; int nblk_atoms;
    float* slab_mol; int slab_mol_stride; int nblk_mol;
    // block-row readout (mkgnn_readout_blocks_*): `pre` holds propagate(W1 sim) WITHOUT the bias, added where it is read
    const float* pre_bias;         // b1 (or null: pre includes it)
    float* gsum;                   // [nmol, HP] or null.  Forward (pool kernel): `pre` is OVERWRITTEN with the gate keep * swish'(pre + b1)
                                   // the backward multiplies by, gsum receives its per-molecule sums; backward (mol kernel): db1 from it
};

__device__ __forceinline__ float sigmoid_f(float p) { return 1.f / (1.f + expf(-p)); }

// A [rows, cols] row-major weight matrix -> LDS image [rows_pad][ld] (zero outside), eight loads in flight per thread.
// (Written as "for (i ...) lds[i] = ok ? w[...] : 0" the compiler keeps one conditional load in flight at a time:
// 16 dependent round trips ahead of readout_pre_kernel's first tile, ~10 of its 21 us.)
template <int NTHREADS>
__device__ __forceinline__ void weights_to_lds(float* lds, int total, int ld, const float* w, int rows, int cols, int tid) {
    for (int base = 0; base < total; base += NTHREADS * 8) {
        float tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + tid + NTHREADS * k;
            const int ic = i < total ? i : total - 1;
            const int r = ic / ld, c = ic - r * ld;
            const bool ok = r < rows && c < cols;
            const float v = w[ok ? r * cols + c : 0];          // unconditional load, masked after
            tmp[k] = ok ? v : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + tid + NTHREADS * k;
            if (i < total) lds[i] = tmp[k];
        }
    }
}

// ------------------------------------------------------------------ forward: pre = h W1^T + b1 ----
template <int NT, int NJ>
__global__ void __launch_bounds__(256) readout_pre_kernel(ReadoutArgs a) {
    constexpr int HP = 16 * NT, FP = 16 * NJ, LDW = FP + 4;
    __shared__ __attribute__((aligned(16))) float w1s[HP * LDW];
    weights_to_lds<256>(w1s, HP * LDW, LDW, a.w1, a.H, a.F, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = (a.b1 && 16 * t + r < a.H) ? a.b1[16 * t + r] : 0.f;
    const int64_t ntiles = (a.n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        int64_t atom = tile * 16 + r;
        if (atom >= a.n) atom = a.n - 1;
        const float* row = a.h + atom * a.hs;
        f32x4 v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = 16 * j + 4 * q;
            v[j] = *(const f32x4*)(row + (col < a.F ? col : 0));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = 16 * j + 4 * q;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[j][c] = (col + c < a.F) ? v[j][c] : 0.f;   // row padding may hold anything
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 w = *(const f32x4*)&w1s[(16 * t + r) * LDW + 16 * j + 4 * q];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][c], w[c], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t ao = tile * 16 + 4 * q + i;
                if (ao < a.n) a.pre[ao * HP + 16 * t + r] = acc[t][i] + bias[t];
            }
        }
    }
}

// -------------------------------------- forward: pooled_g = sum keep*swish(pre); out_g = W2 pooled_g + |g| b2 ----
__global__ void __launch_bounds__(256) readout_pool_kernel(ReadoutArgs a, int HP) {
    __shared__ float w2s[64 * 65];
    __shared__ float scr[4][64];
    const int H = a.H, G = a.G;
    weights_to_lds<256>(w2s, G * (H + 1), H + 1, a.w2, G, H, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = 64 / HP, c = lane & (HP - 1), gid = lane / HP;
    const bool kc = a.keep && c < H;
    const float pb = (a.pre_bias && c < H) ? a.pre_bias[c] : 0.f;
    for (int64_t mol = (int64_t)blockIdx.x * 4 + wave; mol < a.nmol; mol += (int64_t)gridDim.x * 4) {
        const int lo = a.mol_ptr[mol], hi = a.mol_ptr[mol + 1];
        float s = 0.f, gs = 0.f;
        for (int at0 = lo + gid; at0 < hi; at0 += 4 * groups) {
            float p[4], k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int at = at0 + u * groups;
                const int atc = at < hi ? at : hi - 1;
                p[u] = a.pre[(int64_t)atc * HP + c] + pb;
                k[u] = kc ? a.keep[(int64_t)atc * H + c] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sg = sigmoid_f(p[u]);
                const float v = p[u] * sg * k[u];
                if (at0 + u * groups < hi) {
                    s += v;
                    if (a.gsum) {                                // (wave-uniform) training: leave the backward's gate in place of pre
                        const float gate = c < H ? k[u] * (sg * fmaf(p[u], 1.f - sg, 1.f)) : 0.f;
                        a.pre[(int64_t)(at0 + u * groups) * HP + c] = gate;
                        gs += gate;
                    }
                }
            }
        }
        if (groups == 2) { s += __shfl_xor(s, 32, 64); gs += __shfl_xor(gs, 32, 64); }
        else if (groups == 4) { s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64); gs += __shfl_xor(gs, 16, 64); gs += __shfl_xor(gs, 32, 64); }
        if (gid == 0) { a.pooled[mol * HP + c] = s; scr[wave][c] = s; if (a.gsum) a.gsum[mol * HP + c] = gs; }
        __builtin_amdgcn_wave_barrier();
        if (lane < G) {
            float z = 0.f;
            for (int cc = 0; cc < H; ++cc) z = fmaf(w2s[lane * (H + 1) + cc], scr[wave][cc], z);
            if (a.b2) z = fmaf((float)(hi - lo), a.b2[lane], z);
            a.out[mol * a.os + lane] = z;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------- backward, per molecule: dA_g = dz_g W2; partial dW2 = dz^T pooled, db2 = sum |g| dz_g ----
__global__ void __launch_bounds__(256) readout_bwd_mol_kernel(ReadoutArgs a, int HP) {
    constexpr int MC = 16;
    __shared__ float w2s[64 * 65];
    __shared__ float dzs[MC][64];
    __shared__ float As[MC][64];
    __shared__ float nat[MC];
    const int H = a.H, G = a.G, tid = threadIdx.x;
    weights_to_lds<256>(w2s, G * (H + 1), H + 1, a.w2, G, H, tid);
    const int64_t per = (a.nmol + gridDim.x - 1) / gridDim.x;
    const int64_t m_lo = per * blockIdx.x, m_hi = (m_lo + per < a.nmol) ? m_lo + per : a.nmol;
    float accw[16];
    int po[16], pc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        accw[k] = 0.f;
        int p = tid + 256 * k;
        if (p >= G * H) p = 0;
        po[k] = p / H; pc[k] = p - po[k] * H;
    }
    float accb = 0.f, accb1 = 0.f;
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += MC) {
        __syncthreads();
        {   // MC * 64 = 4 * 256 entries: all loads first (unconditional, clamped), then the LDS stores
            float dz[4], av[4];
            int p0[4], p1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + 256 * k, m = i >> 6, o = i & 63;
                const int64_t mc = m0 + m < m_hi ? m0 + m : m_hi - 1;
                dz[k] = a.gout[mc * a.gos + (o < G ? o : 0)];
                av[k] = a.pooled[mc * HP + (o < H ? o : 0)];
                p0[k] = a.mol_ptr[mc]; p1[k] = a.mol_ptr[mc + 1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + 256 * k, m = i >> 6, o = i & 63;
                const bool ok = m0 + m < m_hi;
                dzs[m][o] = (ok && o < G) ? dz[k] : 0.f;
                As[m][o] = (ok && o < H) ? av[k] : 0.f;
                if (o == 0) nat[m] = ok ? (float)(p1[k] - p0[k]) : 0.f;
            }
        }
        __syncthreads();
        for (int m = 0; m < MC; ++m) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (256 * k < G * H) accw[k] = fmaf(dzs[m][po[k]], As[m][pc[k]], accw[k]);      // (uniform: G * H = 1024 uses 4 of 16)
            if (tid < G) accb = fmaf(nat[m], dzs[m][tid], accb);
        }
        for (int i = tid; i < MC * HP; i += 256) {
            const int m = i / HP, c = i - m * HP;
            if (m0 + m < m_hi) {
                float v = 0.f;
                if (c < H)
                    for (int o = 0; o < G; ++o) v = fmaf(dzs[m][o], w2s[o * (H + 1) + c], v);
                a.dA[(m0 + m) * HP + c] = v;
                // block-row readout: db1 = sum_n dpre[n] = sum_mol dA[mol] * (sum of the molecule's gates); a thread's column
                // c = tid % HP is fixed (HP divides 256)
                if (a.gsum && c < H) accb1 = fmaf(v, a.gsum[(m0 + m) * HP + c], accb1);
            }
        }
    }
    float* slab = a.slab_mol + (int64_t)blockIdx.x * a.slab_mol_stride;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int p = tid + 256 * k; if (p < G * H) slab[p] = accw[k]; }
    if (tid < G) slab[G * H + tid] = accb;
    if (a.gsum) {                                            // db1 partial of this block: the 256 / HP thread groups in a fixed order
        __syncthreads();
        float* red = &dzs[0][0];                             // (MC * 64 >= 256 floats)
        red[tid] = accb1;
        __syncthreads();
        if (tid < HP) {
            float t = 0.f;
            for (int k = 0; k < 256 / HP; ++k) t += red[k * HP + tid];
            slab[G * H + G + tid] = t;
        }
    }
}

// ----------------------------------- backward, per atom tile: dpre, dh = dpre W1, dW1 += dpre^T h, db1 += dpre ----
template <int NT, int NU>
__global__ void __launch_bounds__(512) readout_bwd_atoms_kernel(ReadoutArgs a) {
    constexpr int NJ = 4 * NU, HP = 16 * NT, FP = 16 * NJ, LDW = FP + 4, LDP = HP + 4;
    constexpr int NW = NT == 4 ? 4 : 8;                               // two waves per SIMD where the 64 KB of static LDS allow it
    __shared__ __attribute__((aligned(16))) float w1s[HP * LDW];      // [hidden][feature]; reused for the block reduction
    __shared__ __attribute__((aligned(16))) float dps[NW][16 * LDP];  // per wave: dpre tile [atom][hidden]
    weights_to_lds<64 * NW>(w1s, HP * LDW, LDW, a.w1, a.H, a.F, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    float* dpw = dps[wave];
    f32x4 accw[NT][NJ];
    f32x4 colsum[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        colsum[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NJ; ++t) accw[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t ntiles = (a.n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
        // operand of the weight product: h rows 4 s + q, columns 64 u + 4 r .. + 3 (issued first: longest latency)
        f32x4 hv[4][NU];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int64_t ra = tile * 16 + 4 * s + q;
            if (ra >= a.n) ra = a.n - 1;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int col = 64 * u + 4 * r;
                hv[s][u] = *(const f32x4*)(a.h + ra * a.hs + (col < a.F ? col : 0));
            }
        }
        const int64_t atom = tile * 16 + r;
        const bool valid = atom < a.n;
        const int64_t atomc = valid ? atom : a.n - 1;
        const int mol = a.atom_mol[atomc];
        f32x4 dp[NT];
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) {
            const int c0 = 16 * jj + 4 * q;
            const f32x4 p = *(const f32x4*)(a.pre + atomc * HP + c0);
            const f32x4 g = *(const f32x4*)(a.dA + (int64_t)mol * HP + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sg = sigmoid_f(p[c]);
                float d = g[c] * (sg * fmaf(p[c], 1.f - sg, 1.f));
                if (a.keep) d *= (c0 + c < a.H) ? a.keep[atomc * a.H + c0 + c] : 0.f;
                dp[jj][c] = valid ? d : 0.f;
            }
            colsum[jj] += dp[jj];
            *(f32x4*)&dpw[r * LDP + c0] = dp[jj];
        }
        // dh tile = dpre [16 x HP] . W1 [HP x FP]
        if (a.gh) {
            f32x4 acch[NJ];
#pragma unroll
            for (int t = 0; t < NJ; ++t) acch[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* wrow = &w1s[(16 * jj + 4 * q + c) * LDW + r];
#pragma unroll
                    for (int t = 0; t < NJ; ++t)
                        acch[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[jj][c], wrow[16 * t], acch[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int col = 16 * t + r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t ao = tile * 16 + 4 * q + i;
                    if (ao < a.n && col < a.F) a.gh[ao * a.ghs + col] = acch[t][i];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // dW1 += dpre^T [HP x 16 atoms] . h [16 atoms x FP]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int col = 64 * u + 4 * r;
#pragma unroll
                for (int c = 0; c < 4; ++c) hv[s][u][c] = (col + c < a.F) ? hv[s][u][c] : 0.f;
            }
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const float at = dpw[(4 * s + q) * LDP + 16 * mt + r];
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        accw[mt][4 * u + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, hv[s][u][c], accw[mt][4 * u + c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // block reduction in wave order, then one slab row per block
    float* red = w1s;
#pragma unroll
    for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = colsum[jj][c];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            colsum[jj][c] = v;
        }
    // LDS image in lane order -- value k of lane l at red[k * 64 + l]: conflict-free.  (Indexed by (hidden, feature) the
    // lanes of an access were 512 bytes apart: 8-way bank conflicts.)  Cycle stamps of this kernel at batch 4096: weight
    // copy 3.5 k, three tiles of 14-20 k each, a fourth for one wave in eight (6412 tiles over 2048 waves) that the
    // rest of its block waits for at the barrier below, reduction + slab 5 k.
    // The (hidden, feature) mapping is applied once, on the way to the slab.
    constexpr int NV = NT * NJ * 4;                    // weight-gradient values per lane
    for (int w = 0; w < NW; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int t = 0; t < NJ; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = ((mt * NJ + t) * 4 + i) * 64 + lane;
                        red[idx] = (w == 0) ? accw[mt][t][i] : red[idx] + accw[mt][t][i];
                    }
            if (r == 0) {
#pragma unroll
                for (int jj = 0; jj < NT; ++jj)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int idx = NV * 64 + 16 * jj + 4 * q + c;
                        red[idx] = (w == 0) ? colsum[jj][c] : red[idx] + colsum[jj][c];
                    }
            }
        }
    }
    __syncthreads();
    float* slab = a.slab_atoms + (int64_t)blockIdx.x * a.slab_atoms_stride;
    for (int e = threadIdx.x; e < NV * 64; e += 64 * NW) {
        const int k = e >> 6, ln = e & 63, rr = ln & 15, qq = ln >> 4;
        const int i = k & 3, t = (k >> 2) % NJ, mt = (k >> 2) / NJ;
        const int hid = 16 * mt + 4 * qq + i, feat = 64 * (t >> 2) + 4 * rr + (t & 3);
        slab[hid * FP + feat] = red[e];
    }
    for (int e = threadIdx.x; e < HP; e += 64 * NW) slab[HP * FP + e] = red[NV * 64 + e];
}

// ------------------------------------------------ fixed-order sum of per-block slabs into the parameters ----
struct SlabSeg {
    const float* src; int stride; int count;    // count slabs, `stride` floats apart
    int src_cols, dst_rows, dst_cols;           // 2-D window [dst_rows, dst_cols] of a [*, src_cols] slab image
    float* dst;
    int blk_start;
    int dst_stride;                             // row stride of dst (0: dst_cols -- contiguous)
};
struct SlabReduceArgs {
    SlabSeg seg[12]; int nseg;
    float drop_p; int64_t* rng; int64_t* rng_used;      // the fused tail: the head's dropout generator advances here, once per step
};

// (blk == 0: none) mkgnn_tail_args.defer_reduce.  Per DEVICE, not per thread: the forward that leaves it runs on the caller's
// thread, the backward that takes it on autograd's.  (The header's rule -- one host thread per device inside these calls at a
// time -- is what orders the two; the mutex only keeps the slot itself whole.)
struct PendingReduce { SlabReduceArgs r; int blk; };
static PendingReduce g_pending_reduce_dev[16];
static std::mutex g_pending_reduce_mutex;
static PendingReduce* pending_reduce_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    return &g_pending_reduce_dev[dev];
}

__global__ void __launch_bounds__(256) slab_reduce_kernel(SlabReduceArgs a) {
    __shared__ float part[8][32];
    int si = 0;
    for (int s = 1; s < a.nseg; ++s) if ((int)blockIdx.x >= a.seg[s].blk_start) si = s;
    const SlabSeg g = a.seg[si];
    const int e = (blockIdx.x - g.blk_start) * 32 + (threadIdx.x & 31), p = threadIdx.x >> 5;
    const int total = g.dst_rows * g.dst_cols;
    const int ec = e < total ? e : total - 1;
    const int row = ec / g.dst_cols, col = ec - row * g.dst_cols;
    const float* src = g.src + row * g.src_cols + col;
    const int per = (g.count + 7) / 8;
    const int b0 = p * per, b1 = (b0 + per < g.count) ? b0 + per : g.count;
    float s = 0.f;
    for (int b = b0; b < b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + u < b1 ? b + u : b1 - 1) * g.stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (b + u < b1) s += v[u];
    }
    part[p][threadIdx.x & 31] = s;
    __syncthreads();
    if (p == 0 && e < total) {
        float t = part[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += part[k][threadIdx.x];
        g.dst[g.dst_stride ? (size_t)row * g.dst_stride + col : (size_t)e] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.drop_p > 0.f && a.rng) {      // (as behind mkgnn_bce_head_fused)
        const int64_t seed = a.rng[0], offset = a.rng[1];
        a.rng_used[0] = seed; a.rng_used[1] = offset;
        a.rng[1] = offset + 1;
    }
}

// ---------------------------------------------------------------------- block-row readout (round 3) ----
// d loss / d z = propagate^T(dpre), dpre[t] = dA[mol(t)] * gate[t] taken on the fly (gate = keep * swish'(pre + b1), left
// in place of pre by the forward's pool kernel): row n of dz is the sum of dpre over n's neighbours t (the CSR of the edges by
// source) -- the [N x H] dpre array is never written or read.  CPR lanes per row (16-byte chunks of the HP-wide rows),
// 256 / CPR rows per pass; the first four neighbours' loads are all in flight at once (atoms have at most four neighbours
// but for a handful: the rest of such a row follows serially).
struct DzArgs {
    const float* dA; const float* gate; const int32_t* atom_mol;
    const int32_t* rowptr; const int32_t* col; int64_t n;
    float* dz;
};
constexpr int DZ_BLOCKS = 2048;

template <int CPR>
__global__ void __launch_bounds__(256) readout_dz_gather_kernel(DzArgs a) {
    constexpr int HP = 4 * CPR, RPB = 256 / CPR;
    const int tid = threadIdx.x, l = tid % CPR, rs = tid / CPR, c0 = 4 * l;
    for (int64_t r = (int64_t)blockIdx.x * RPB + rs; r < a.n; r += (int64_t)gridDim.x * RPB) {
        const int e0 = a.rowptr[r], e1 = a.rowptr[r + 1];
        int64_t t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = e0 + j < e1 ? (int64_t)a.col[e0 + j] : r;         // (clamped loads, masked below)
        int m[4];
        f32x4 g4[4], d4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = a.atom_mol[t[j]]; g4[j] = *(const f32x4*)(a.gate + t[j] * HP + c0); }
#pragma unroll
        for (int j = 0; j < 4; ++j) d4[j] = *(const f32x4*)(a.dA + (int64_t)m[j] * HP + c0);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) if (e0 + j < e1) acc += d4[j] * g4[j];
        for (int e = e0 + 4; e < e1; ++e) {                                                  // (more than four neighbours)
            const int64_t tt = a.col[e];
            acc += *(const f32x4*)(a.dA + (int64_t)a.atom_mol[tt] * HP + c0) * *(const f32x4*)(a.gate + tt * HP + c0);
        }
        *(f32x4*)(a.dz + r * HP + c0) = acc;
    }
}

// The last kernel convolution's output sim goes nowhere but into  h = propagate(sim)  and  pre = W1 h + b1  (reference
// KernelLayer.py:119-123, MolKGNNNet.py:144-146).  Both are linear, and row n of sim is non-zero only in the column block
// of atom n's degree: project first, z[n] = W1[:, block(n)] sim[n, block(n)]  (H numbers from L_d), then propagate the
// H-wide rows instead of the K-wide ones, pre = propagate(z) (+ b1 where it is read).  Same sums, re-associated.  Removes a
// 45 MB dense h, its 45 MB gradient and the [N x K] x [K x H] tile products from the step.
struct BlockProjArgs {
    const float* sim; int64_t ss; int64_t n;
    const float* w1; int H, K, HP;
    int off[MKGNN_MAX_DEGREE], L[MKGNN_MAX_DEGREE];
    float* z;                                   // forward: [n, HP]
    // backward
    const float* dz;                            // [n, HP] = propagate^T(dpre)
    const int64_t* sel[MKGNN_MAX_DEGREE]; int64_t cnt[MKGNN_MAX_DEGREE];
    float* dsim; int64_t dss;                   // [n, K] block rows (only every atom's own block is written)
    float* slab; int slab_stride; int FP;       // per block: dW1 image [HP][FP]
};

// Both kernels work on TILES OF ONE DEGREE BUCKET (16 atoms of degree d in selected_index order, so the block [off, off + L)
// is the same for the whole tile) on the fp32 matrix cores; a block's four waves take four consecutive tiles of one bucket
// and share that degree's slice of W1 in LDS ([HP][LP], rows 16-byte aligned, zero beyond H / L).
//
// Lane layout of a tile (r = lane & 15, q = lane >> 4), as in readout_pre_kernel: lane (r, q) holds, of atom r's row, the four
// consecutive columns 16 j + 4 q .. + 3 of chunk j -- an MFMA's k index may be any permutation as long as both operands use
// the same one.
struct BlockTile { int di; int64_t t; int64_t cnt; int L, off, nj; };
__device__ __forceinline__ bool block_tile_of(const BlockProjArgs& a, int64_t blk, int wave, BlockTile& T) {
    // blocks are numbered bucket by bucket: bucket d has ceil(ceil(cnt_d / 16) / 4) of them
    int64_t b0 = 0;
    for (int di = 0; di < MKGNN_MAX_DEGREE; ++di) {
        const int64_t tiles = (a.cnt[di] + 15) / 16, nb = a.L[di] > 0 ? (tiles + 3) / 4 : 0;
        if (blk < b0 + nb) {
            T.di = di; T.t = (blk - b0) * 4 + wave; T.cnt = a.cnt[di]; T.L = a.L[di]; T.off = a.off[di]; T.nj = (a.L[di] + 15) / 16;
            return true;
        }
        b0 += nb;
    }
    return false;
}
// four consecutive floats of a row whose alignment (in floats, mod 4) is wave-uniform: one, two or four loads
__device__ __forceinline__ f32x4 load4_at(const float* p, int align4) {
    f32x4 v;
    if (align4 == 0) v = *(const f32x4*)p;
    else if (align4 == 2) { const float2 lo = *(const float2*)p, hi = *(const float2*)(p + 2); v = f32x4{lo.x, lo.y, hi.x, hi.y}; }
    else v = f32x4{p[0], p[1], p[2], p[3]};
    return v;
}
// this degree's slice of W1 -> LDS [HP][LP] (LP = 16 nj + 4), zero beyond H and L
__device__ __forceinline__ void w1_block_to_lds(float* w1s, const BlockProjArgs& a, int HP, int L, int off, int LP, int tid) {
    for (int i = tid; i < HP * LP; i += 256) {
        const int hid = i / LP, l = i - hid * LP;
        w1s[i] = (hid < a.H && l < L) ? a.w1[(size_t)hid * a.K + off + l] : 0.f;
    }
}

// z[n] = W1[:, block(n)] sim[n, block(n)]  for the atoms of the degree buckets: [16 atoms x L] . [L x HP] per tile
template <int NT>
__global__ void __launch_bounds__(256) block_project_mfma_kernel(BlockProjArgs a) {
    constexpr int HP = 16 * NT, LZ = HP + 4;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    BlockTile T;
    if (!block_tile_of(a, blockIdx.x, wave, T)) return;
    const int LP = 16 * T.nj + 4;
    float* const w1s = lds;
    float* const zt = lds + HP * 68 + wave * 16 * LZ;             // this wave's [16][LZ] transpose image
    w1_block_to_lds(w1s, a, HP, T.L, T.off, LP, tid);
    // the tile's rows while the weights arrive: lane (r, q) <- atom r, columns 16 j + 4 q .. + 3
    const int64_t p = T.t * 16 + r, pc = p < T.cnt ? p : T.cnt - 1;
    const bool tile_ok = T.t * 16 < T.cnt;
    const int64_t id = tile_ok ? a.sel[T.di][pc] : 0;
    const float* row = a.sim + id * a.ss + T.off;
    const int al = T.off & 3;                                      // (row bases are 16-byte aligned: the block's offset decides)
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = 16 * j + 4 * q;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < T.nj && tile_ok) {
            if (col + 3 < T.L) v[j] = load4_at(row + col, al);
            else {                                                 // the block's last, partial chunk: element by element
#pragma unroll
                for (int c = 0; c < 4; ++c) v[j][c] = col + c < T.L ? row[col + c] : 0.f;
            }
        }
    }
    __syncthreads();
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < T.nj) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 w = *(const f32x4*)&w1s[(16 * t + r) * LP + 16 * j + 4 * q];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j][c], w[c], acc[t], 0, 0, 0);
            }
        }
    }
    // acc[t][i] = z[atom 4 q + i][16 t + r]: through the wave's LDS image to whole 16-byte chunks of the z rows
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) zt[(4 * q + i) * LZ + 16 * t + r] = acc[t][i];
    __builtin_amdgcn_wave_barrier();
    constexpr int CPRZ = HP / 4;                                   // chunks per z row (8 or 16)
#pragma unroll
    for (int c0 = 0; c0 < 16 * CPRZ; c0 += 64) {
        const int cidx = c0 + lane, at = cidx / CPRZ, ch = cidx - at * CPRZ;
        const int64_t ida = __shfl(id, at, 64);                    // (lane `at` holds atom at's id: r = at, q = 0)
        if (T.t * 16 + at < T.cnt) *(f32x4*)(a.z + ida * HP + 4 * ch) = *(const f32x4*)&zt[at * LZ + 4 * ch];
    }
}

// Backward per tile:  dsim[n, block] = dz[n] W1[:, block]  ([16 x HP] . [HP x L]),  dW1[:, block] += dz^T sim  ([HP x 16] . [16 x L],
// accumulators in registers over the block's tiles, then block -> slab -> fixed-order reduction, one slab image per block
// holding only its degree's columns).
template <int NT>
__global__ void __launch_bounds__(256) block_project_bwd_mfma_kernel(BlockProjArgs a, int tiles_per_wave) {
    constexpr int HP = 16 * NT, LZ = HP + 4;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    // blocks bucket by bucket; a block takes 4 * tiles_per_wave consecutive tiles of its bucket
    int di = -1; int64_t blk_in = 0;
    {
        int64_t b0 = 0;
        for (int k = 0; k < MKGNN_MAX_DEGREE; ++k) {
            const int64_t tiles = (a.cnt[k] + 15) / 16, per = 4 * (int64_t)tiles_per_wave;
            const int64_t nb = a.L[k] > 0 ? (tiles + per - 1) / per : 0;
            if (di < 0 && (int64_t)blockIdx.x < b0 + nb) { di = k; blk_in = blockIdx.x - b0; }
            b0 += nb;
        }
    }
    if (di < 0) return;
    const int L = a.L[di], off = a.off[di], nj = (L + 15) / 16, LP = 16 * nj + 4;
    const int64_t cnt = a.cnt[di];
    float* const w1s = lds;                                        // [HP][LP]
    float* const dzs = lds + HP * 68 + wave * (16 * LZ + 16 * 68); // per wave: dz tile [16][LZ] | sim tile [16][68]
    float* const sms = dzs + 16 * LZ;
    w1_block_to_lds(w1s, a, HP, L, off, LP, tid);
    __syncthreads();
    const int al = off & 3;
    f32x4 accw[NT][4];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) accw[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int64_t tile = (blk_in * 4 + wave) * tiles_per_wave + it;
        if (tile * 16 >= cnt) break;                               // (wave-uniform)
        const int64_t p = tile * 16 + r, pc = p < cnt ? p : cnt - 1;
        const int64_t id = a.sel[di][pc];
        const bool valid = p < cnt;
        // dz rows as the A operand of the dsim product (k = hidden unit 16 jh + 4 q + c); sim blocks for the dW1 product
        f32x4 dzv[NT], sv[4];
#pragma unroll
        for (int jh = 0; jh < NT; ++jh) dzv[jh] = *(const f32x4*)(a.dz + id * HP + 16 * jh + 4 * q);
        const float* row = a.sim + id * a.ss + off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 16 * j + 4 * q;
            sv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < nj) {
                if (col + 3 < L) sv[j] = load4_at(row + col, al);
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) sv[j][c] = col + c < L ? row[col + c] : 0.f;
                }
            }
        }
#pragma unroll
        for (int jh = 0; jh < NT; ++jh) {
            if (!valid) dzv[jh] = f32x4{0.f, 0.f, 0.f, 0.f};       // (a padding atom of the last tile contributes nothing)
            *(f32x4*)&dzs[r * LZ + 16 * jh + 4 * q] = dzv[jh];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (j < nj) *(f32x4*)&sms[r * 68 + 16 * j + 4 * q] = sv[j];
        __builtin_amdgcn_wave_barrier();
        // ---- dsim tile
        if (a.dsim) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nj) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jh = 0; jh < NT; ++jh)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv[jh][c], w1s[(16 * jh + 4 * q + c) * LP + 16 * t + r], acc, 0, 0, 0);
                    const int col = 16 * t + r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int at = 4 * q + i;
                        const int64_t ida = __shfl(id, at, 64);
                        if (tile * 16 + at < cnt && col < L) a.dsim[ida * a.dss + off + col] = acc[i];
                    }
                }
            }
        }
        // ---- dW1 += dz^T . sim  (k = atom 4 s + q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float av[NT], bv[4];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) av[mt] = dzs[(4 * s4 + q) * LZ + 16 * mt + r];
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[t] = t < nj ? sms[(4 * s4 + q) * 68 + 16 * t + r] : 0.f;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < nj) accw[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], accw[mt][t], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- block reduction of the four waves' accumulators (fixed order), one hidden tile at a time, then this block's slab:
    // [HP][FP], only its degree's columns
    float* const red = lds + HP * 68;                              // reuses the waves' tile images: [4 waves][4 column tiles][256]
    float* slab = a.slab + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wave * 4 + t) * 256 + i * 64 + lane] = accw[mt][t][i];
        __syncthreads();
        for (int e = tid; e < 4 * 256; e += 256) {
            const int t = e >> 8, i = (e & 255) >> 6, ln = e & 63;
            if (t < nj) {
                const float sum = (red[(0 * 4 + t) * 256 + (e & 255)] + red[(1 * 4 + t) * 256 + (e & 255)]) +
                                  (red[(2 * 4 + t) * 256 + (e & 255)] + red[(3 * 4 + t) * 256 + (e & 255)]);
                // accumulator element (i, lane): row (hidden) 16 mt + 4 (ln >> 4) + i, column 16 t + (ln & 15)
                const int hid = 16 * mt + 4 * (ln >> 4) + i, col = 16 * t + (ln & 15);
                if (col < L) slab[(size_t)hid * a.FP + off + col] = sum;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ batch norm ----
// Statistics-only companion of a batch norm (mkgnn_bn_stats): the reference runs edge_batch_norm(data.edge_attr) in every
// forward (MolKGNNNet.py:116) although its output never reaches the kernel convolution; what remains of the
// call is its side effect in training mode -- running_mean / running_var / num_batches_tracked move.  The rows are summed
// by extra blocks of the node batch norm's own two launches (no launch of their own in a step), or by
// mkgnn_batchnorm_update_stats alone.  part: [3][nblk][C] = column sums | squares about the block's own means | counted rows.
#ifndef MKGNN_BN_BLOCKS                  // (A/B builds)
#define MKGNN_BN_BLOCKS 256
#endif
constexpr int BN_MAIN_BLOCKS = MKGNN_BN_BLOCKS;        // (= BN_BLOCKS: the grid of the batch norm's own passes)
constexpr int BN_SIDE_BLOCKS = 1024;                   // companion blocks at most (one loop trip per pass each where that is enough)
struct BnSide {
    const float* x; int64_t xs; int64_t n; int C, CL, nblk;
    int flat;                                          // contiguous narrow rows: the 16-byte form (bn_side_block_stats_flat)
    float *running_mean, *running_var; float momentum;
    int64_t* nbt;
    const int64_t* key; const int64_t* key_limit;      // both or neither: row r counts iff key[r] < *key_limit
    float* part;
};

struct BnArgs {
    const float* x; int64_t xs; int64_t n; int C;
    const float *weight, *bias;
    float *running_mean, *running_var;
    float momentum, eps; int training;
    float* out; int64_t os;
    float *save_mean, *save_invstd;
    float* part1; float* part2;    // [nblk, C] each
    // backward
    const float* gout; int64_t gos;
    float* gx; int64_t gxs;
    float *gweight, *gbias;
    float* inv_out;                // forward: 1 / max(|out row|, eps) for the convolution that reads out next (17 <= C <= 32)
    int split_out;                 // ... and the rows themselves written PRE-SPLIT for it (kgnn_split.h; MKGNN_BN_SPLIT_ROWS)
    int touch_first;               // blocks from here on of the statistics launch read `touch` and throw it away (mkgnn_touch_hint)
    TouchArgs touch;
    int64_t* nbt;                  // forward, training: BatchNorm1d.num_batches_tracked, incremented
    const int64_t* nvalid;         // device scalar or null: only rows [0, *nvalid) enter the batch statistics (padded batches)
    BnSide side;                   // statistics-only companion (blocks gridDim.x - side.nblk .. of the same launches); nblk = 0: none
};

// blocks that work on the batch norm's own rows: the grid, less the companion's extra blocks (BnSide) behind them
__device__ __forceinline__ int bn_nblk() { return (int)gridDim.x < BN_MAIN_BLOCKS ? (int)gridDim.x : BN_MAIN_BLOCKS; }

// rows that count for the statistics: all of them, or the leading *nvalid (the rest is padding: normalised like any
// row, excluded from every sum)
__device__ __forceinline__ int64_t bn_valid(const BnArgs& a) {
    if (!a.nvalid) return a.n;
    const int64_t v = *a.nvalid;
    return v < 1 ? 1 : (v < a.n ? v : a.n);
}

// rows of this block: [lo, hi)
__device__ __forceinline__ void bn_rows(const BnArgs& a, int64_t& lo, int64_t& hi) {
    const int64_t per = (a.n + bn_nblk() - 1) / bn_nblk();
    lo = per * blockIdx.x;
    hi = lo + per < a.n ? lo + per : a.n;
    if (lo > hi) lo = hi;
}

// ---- round 6: statistics in ONE launch, 16-byte row passes ------------------------------------------------------------------
// Rounds 1-5 took the batch statistics in two launches (column sums; then, with the batch mean known, centred squares) and
// applied them in a third; every row pass was 4-byte loads, eight in flight per thread -- 8.6 + 8.9 + 14.2 us for an 11.5 MB
// tensor at batch 4096, bound by load latency times loop trips.  Now a block sums its share of the counted rows (pass A), takes
// ITS OWN mean, and sums the squares about that (pass B: the rows the block has just pulled into L2).  The batch statistics
// follow exactly from the block triples (count_b, sum_b, M2_b):
//     mean = (sum_b sum_b) / n,     M2 = sum_b [ M2_b + count_b (sum_b / count_b - mean)^2 ]          (Chan, Golub, LeVeque 1979)
// evaluated in a fixed order by every block of the apply launch -- centred sums throughout, like the two-pass form over the
// whole batch it replaces; one launch less, and the mean itself is bit for bit the old one (the same partial sums).  Rows that
// are 16-byte aligned with C a multiple of 4 are read as float4, BN_U rows in flight per thread.
constexpr int BN_U = 16;                               // float4 row loads in flight per thread (statistics passes)
constexpr int BN_UA = 8;                               // ... in the passes that also write rows

__device__ __forceinline__ float bn_shfl_xor(float v, int o) { return __shfl_xor(v, o, 64); }
__device__ __forceinline__ f32x4 bn_shfl_xor(f32x4 v, int o) {
    return f32x4{__shfl_xor(v[0], o, 64), __shfl_xor(v[1], o, 64), __shfl_xor(v[2], o, 64), __shfl_xor(v[3], o, 64)};
}
__device__ __forceinline__ void bn_zero(float& v) { v = 0.f; }
__device__ __forceinline__ void bn_zero(f32x4& v) { v = f32x4{0.f, 0.f, 0.f, 0.f}; }

// Sum of v over the threads of the block that share threadIdx.x % LW (LW a power of two, 1 .. 256), in a fixed order: an xor
// tree over the lanes of a wave, then the waves (or row lanes) in ascending order.  Valid in threads < LW.  shv: 256 V's.
template <typename V> __device__ __forceinline__ V bn_reduce_rows(V v, int LW, V* shv) {
    const int t = threadIdx.x;
    for (int o = 32; o >= LW; o >>= 1) v = v + bn_shfl_xor(v, o);
    __syncthreads();                                     // (shv may still be read from an earlier reduction)
    int parts;
    if (LW <= 64) {
        if ((t & 63) < LW) shv[(t >> 6) * LW + (t & 63)] = v;
        parts = 4;
    } else {
        shv[t] = v;
        parts = 256 / LW;
    }
    __syncthreads();
    V r;
    bn_zero(r);
    if (t < LW)
        for (int k = 0; k < parts; ++k) r = r + shv[k * LW + t];
    return r;
}

// rows of block b of nblk, the counted rows [0, nv) dealt in equal runs: [lo, hi)
__device__ __forceinline__ void bn_share(int64_t nv, int nblk, int b, int64_t& lo, int64_t& hi) {
    const int64_t per = (nv + nblk - 1) / nblk;
    lo = per * b;
    hi = lo + per < nv ? lo + per : nv;
    if (lo > hi) lo = hi;
}

// One pass over rows [lo, hi) of x (and g): per-thread partial sums.  MODE 0: s0 += x;  1: s0 += (x - mu)^2;  2: s0 += g,
// s1 += g (x - mu) is.  VEC: thread = (row lane rsub of RS, column group col .. col + 3), float4 loads; else one column c.
// All loads of a trip are issued before anything is consumed (clamped addresses, masked use).
template <int MODE, int U>
__device__ __forceinline__ void bn_pass_vec(const float* x, int64_t xs, const float* g, int64_t gs, int64_t lo, int64_t hi, int rsub, int RS,
                                            int col, f32x4 mu, f32x4 is, f32x4& s0, f32x4& s1) {
    for (int64_t r0 = lo + rsub; r0 < hi; r0 += (int64_t)U * RS) {
        f32x4 v[U], gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r0 + (int64_t)u * RS < hi ? r0 + (int64_t)u * RS : hi - 1;
            v[u] = *(const f32x4*)(x + rr * xs + col);
            if (MODE == 2) gg[u] = *(const f32x4*)(g + rr * gs + col);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r0 + (int64_t)u * RS < hi) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (MODE == 0) s0[e] += v[u][e];
                    if (MODE == 1) { const float d = v[u][e] - mu[e]; s0[e] = fmaf(d, d, s0[e]); }
                    if (MODE == 2) { s0[e] += gg[u][e]; s1[e] = fmaf(gg[u][e], (v[u][e] - mu[e]) * is[e], s1[e]); }
                }
            }
        }
    }
}
template <int MODE>
__device__ __forceinline__ void bn_pass_col(const float* x, int64_t xs, const float* g, int64_t gs, int64_t lo, int64_t hi, int rsub, int RS,
                                            int c, float mu, float is, float& s0, float& s1) {
    for (int64_t r0 = lo + rsub; r0 < hi; r0 += 8 * RS) {
        float v[8], gg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t rr = r0 + u * RS < hi ? r0 + u * RS : hi - 1;
            v[u] = x[rr * xs + c];
            if (MODE == 2) gg[u] = g[rr * gs + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (r0 + u * RS < hi) {
                if (MODE == 0) s0 += v[u];
                if (MODE == 1) { const float d = v[u] - mu; s0 = fmaf(d, d, s0); }
                if (MODE == 2) { s0 += gg[u]; s1 = fmaf(gg[u], (v[u] - mu) * is, s1); }
            }
        }
    }
}

// can the rows of a [*, C] tensor be read as float4?
static inline bool bn_vec_rows(const void* p, int64_t stride, int C) { return p && C % 4 == 0 && stride % 4 == 0 && ((uintptr_t)p & 15) == 0; }

// Forward statistics of block blockIdx.x: part1[b][c] = column sums of its counted rows, part2[b][c] = squares about the
// block's own column means.  sh: 1024 floats.
template <bool VEC>
__device__ __forceinline__ void bn_block_stats(const BnArgs& a, int CL, float* sh) {
    __shared__ float bmean[256];
    const int t = threadIdx.x;
    int64_t lo, hi;
    // the block's share of the rows that COUNT (not of all rows): the partial sums, and with them the statistics, are then
    // bit for bit those of the same batch without its padding rows -- a padded batch (molkgnn_amd.padding) reproduces the
    // unpadded forward exactly, which matters more than it looks: an ulp in x decides thousands of mathematically tied
    // neighbour orders the other way two layers later
    bn_share(bn_valid(a), bn_nblk(), blockIdx.x, lo, hi);
    const float cnt = (float)(hi - lo);
    float* const p1 = a.part1 + (int64_t)blockIdx.x * a.C;
    float* const p2 = a.part2 + (int64_t)blockIdx.x * a.C;
    if constexpr (VEC) {
        const int LW = CL >= 4 ? CL / 4 : 1, g = t % LW, rsub = t / LW, RS = 256 / LW, col = 4 * g;
        const bool act = col < a.C;
        const int cb = act ? col : 0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 s0 = z, s1 = z;
        bn_pass_vec<0, BN_U>(a.x, a.xs, nullptr, 0, lo, hi, rsub, RS, cb, z, z, s0, s1);
        const f32x4 tot = bn_reduce_rows(s0, LW, (f32x4*)sh);
        if (t < LW && act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { p1[col + e] = tot[e]; bmean[col + e] = cnt > 0.f ? tot[e] / cnt : 0.f; }
        }
        __syncthreads();
        const f32x4 mu = {bmean[cb], bmean[cb + 1], bmean[cb + 2], bmean[cb + 3]};
        s0 = z;
        bn_pass_vec<1, BN_U>(a.x, a.xs, nullptr, 0, lo, hi, rsub, RS, cb, mu, z, s0, s1);
        const f32x4 m2 = bn_reduce_rows(s0, LW, (f32x4*)sh);
        if (t < LW && act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) p2[col + e] = m2[e];
        }
    } else {
        const int c = t & (CL - 1), rsub = t / CL, RS = 256 / CL;
        const bool act = c < a.C;
        const int cc = act ? c : 0;
        float s0 = 0.f, s1 = 0.f;
        bn_pass_col<0>(a.x, a.xs, nullptr, 0, lo, hi, rsub, RS, cc, 0.f, 0.f, s0, s1);
        const float tot = bn_reduce_rows(s0, CL, sh);
        if (t < CL && act) { p1[c] = tot; bmean[c] = cnt > 0.f ? tot / cnt : 0.f; }
        __syncthreads();
        const float mu = bmean[cc];
        s0 = 0.f;
        bn_pass_col<1>(a.x, a.xs, nullptr, 0, lo, hi, rsub, RS, cc, mu, 0.f, s0, s1);
        const float m2 = bn_reduce_rows(s0, CL, sh);
        if (t < CL && act) p2[c] = m2;
    }
}

// sum over blocks b of term(b, c) for every column c, identically in every block (fixed order), result in sh_out[c];
// sh: 256 floats
template <typename Term>
__device__ __forceinline__ void bn_total_of(int nblk, int C, int CL, float* sh, float* sh_out, Term&& term) {
    const int c = threadIdx.x & (CL - 1), rsub = threadIdx.x / CL, RS = 256 / CL;
    float s = 0.f;
    if (c < C) {
        // eight terms in flight (every block repeats this sum: serial loads made it the slowest part of the pass)
        float v[8], t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = rsub; b < nblk; b += 8 * RS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = term(b + u * RS < nblk ? b + u * RS : b, c);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (b + u * RS < nblk) t[u] += v[u];
        }
        s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    __syncthreads();
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rsub == 0 && c < C) {
        float t = 0.f;
        for (int k = 0; k < RS; ++k) t += sh[k * CL + c];
        sh_out[c] = t;
    }
    __syncthreads();
}
__device__ __forceinline__ void bn_total(const float* part, int nblk, int C, int CL, float* sh, float* sh_out) {
    bn_total_of(nblk, C, CL, sh, sh_out, [&](int b, int c) { return part[(int64_t)b * C + c]; });
}
// squares about the BATCH mean (mean[c]) out of the blocks' (count, sum, squares about their own means)
__device__ __forceinline__ void bn_total_m2(const float* part1, const float* part2, int nblk, int64_t nv, int C, int CL, const float* mean,
                                            float* sh, float* sh_out) {
    const int64_t per = (nv + nblk - 1) / nblk;          // (one 64-bit division per thread, not one per term)
    const float mu = mean[(threadIdx.x & (CL - 1)) < C ? (threadIdx.x & (CL - 1)) : 0];
    bn_total_of(nblk, C, CL, sh, sh_out, [&](int b, int c) {
        int64_t left = nv - per * b;                     // rows of block b: what bn_share deals it
        left = left < 0 ? 0 : (left > per ? per : left);
        // (both loads unconditional: a load under a condition would be issued -- and waited for -- on its own, term after term)
        const float p1 = part1[(int64_t)b * C + c], p2 = part2[(int64_t)b * C + c];
        const float cnt = (float)left;
        const float d = p1 / fmaxf(cnt, 1.f) - mu;       // (an empty block: p1 = 0, weight 0)
        return fmaf(cnt * d, d, p2);
    });
}

// Companion statistics, block `blk` of s.nblk: plane 0 column sums, plane 1 squares about the block's own column means,
// plane 2 counted rows (every column of the plane holds the count).  Fixed order everywhere.  sh: 1024 floats
__device__ __forceinline__ void bn_side_block_stats(const BnSide& s, int blk, float* sh) {
    __shared__ float smean[256];
    const int CL = s.CL, t = threadIdx.x, c = t & (CL - 1), rsub = t / CL, RS = 256 / CL;
    const bool act = c < s.C;
    const int cc = act ? c : 0;
    int64_t lo, hi;
    bn_share(s.n, s.nblk, blk, lo, hi);
    const int64_t lim = s.key ? *s.key_limit : 0;
    float mu = 0.f, m2 = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float s0 = 0.f, cnt = 0.f;
        for (int64_t r0 = lo + rsub; r0 < hi; r0 += 8 * RS) {
            float v[8];
            bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r0 + u * RS < hi ? r0 + u * RS : hi - 1;
                v[u] = s.x[rr * s.xs + cc];
                ok[u] = r0 + u * RS < hi && (!s.key || s.key[rr] < lim);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (ok[u]) {
                    if (pass == 0) { s0 += v[u]; cnt += 1.f; }
                    else { const float d = v[u] - mu; s0 = fmaf(d, d, s0); }
                }
            }
        }
        const float tot = bn_reduce_rows(s0, CL, sh);
        if (pass == 0) {
            const float n_b = bn_reduce_rows(cnt, CL, sh);          // counts: exact below 2^24 rows per block
            if (t < CL && act) {
                s.part[((size_t)0 * s.nblk + blk) * s.C + c] = tot;
                s.part[((size_t)2 * s.nblk + blk) * s.C + c] = n_b;
                smean[c] = n_b > 0.f ? tot / n_b : 0.f;
            }
            __syncthreads();
            mu = smean[cc];
        } else {
            m2 = tot;
            if (t < CL && act) s.part[((size_t)1 * s.nblk + blk) * s.C + c] = m2;
        }
    }
}

// The same for narrow contiguous rows (x_stride == C <= 8, 16-byte aligned base: the reference's bond rows, [n, 7]): four rows
// are C whole float4s, so a thread takes groups of four rows with C 16-byte loads each, SIDE_G groups in flight -- 4 096 rows per
// block and loop trip where the column-per-thread form above takes 256 (a batch of 4 096 molecules has 216 k bond rows: 53
// blocks of one trip per pass instead of 844, and a final merge over 53 partials instead of 844).
constexpr int SIDE_G = 4;
constexpr int SIDE_FLAT_ROWS = 256 * 4 * SIDE_G;       // rows per block and loop trip
template <int C>
__device__ __forceinline__ void bn_side_block_stats_flat(const BnSide& s, int blk, float* sh) {
    __shared__ float smean[8];
    const int t = threadIdx.x;
    int64_t lo, hi;
    bn_share(s.n / 4, s.nblk, blk, lo, hi);              // in units of four rows: the whole groups; the last n % 4 rows below
    const int64_t lim = s.key ? *s.key_limit : 0;
    float mu[C];
#pragma unroll
    for (int c = 0; c < C; ++c) mu[c] = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float acc[C], cnt = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        for (int64_t q0 = lo + t; q0 - t < hi && lo < hi; q0 += 256 * SIDE_G) {
            f32x4 v[SIDE_G][C];
            bool ok[SIDE_G][4];
#pragma unroll
            for (int g = 0; g < SIDE_G; ++g) {
                const int64_t q = q0 + 256 * g;                       // rows 4 q .. 4 q + 3
                const int64_t qc = q < hi ? q : hi - 1;
#pragma unroll
                for (int c = 0; c < C; ++c) v[g][c] = *(const f32x4*)(s.x + 4 * qc * C + 4 * c);      // (all loads unconditional)
#pragma unroll
                for (int r = 0; r < 4; ++r) ok[g][r] = q < hi && (!s.key || s.key[4 * qc + r] < lim);
            }
#pragma unroll
            for (int g = 0; g < SIDE_G; ++g)
#pragma unroll
                for (int e = 0; e < 4 * C; ++e) {                     // element e of the group: row e / C, column e % C
                    if (ok[g][e / C]) {
                        const float x = v[g][e / 4][e % 4];
                        if (pass == 0) { acc[e % C] += x; if (e % C == 0) cnt += 1.f; }
                        else { const float d = x - mu[e % C]; acc[e % C] = fmaf(d, d, acc[e % C]); }
                    }
                }
        }
        if (blk == s.nblk - 1 && t == 0) {                            // the tensor's last n % 4 rows: a few scalar loads of one thread
            for (int64_t row = s.n / 4 * 4; row < s.n; ++row) {
                if (s.key && !(s.key[row] < lim)) continue;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float x = s.x[row * C + c];
                    if (pass == 0) acc[c] += x;
                    else { const float d = x - mu[c]; acc[c] = fmaf(d, d, acc[c]); }
                }
                if (pass == 0) cnt += 1.f;
            }
        }
        // every thread holds every column: an xor tree over the wave, the four waves in order -- all C + 1 sums behind ONE pair of
        // barriers (one reduction per column cost more than the row pass itself)
        float tot[C], n_b = cnt;
#pragma unroll
        for (int c = 0; c < C; ++c) tot[c] = acc[c];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
            for (int c = 0; c < C; ++c) tot[c] += __shfl_xor(tot[c], o, 64);
            n_b += __shfl_xor(n_b, o, 64);
        }
        __syncthreads();
        if ((t & 63) == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) sh[(t >> 6) * 16 + c] = tot[c];
            sh[(t >> 6) * 16 + 8] = n_b;
        }
        __syncthreads();
        if (t == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) tot[c] = ((sh[c] + sh[16 + c]) + sh[32 + c]) + sh[48 + c];
            n_b = ((sh[8] + sh[24]) + sh[40]) + sh[56];
        }
        if (t == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (pass == 0) {
                    s.part[((size_t)0 * s.nblk + blk) * C + c] = tot[c];
                    s.part[((size_t)2 * s.nblk + blk) * C + c] = n_b;
                    smean[c] = n_b > 0.f ? tot[c] / n_b : 0.f;
                } else {
                    s.part[((size_t)1 * s.nblk + blk) * C + c] = tot[c];
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) mu[c] = smean[c];
        }
    }
}
__device__ __forceinline__ bool bn_side_is_flat(const BnSide& s) { return s.flat != 0; }
__device__ __forceinline__ void bn_side_stats_any(const BnSide& s, int blk, float* sh) {
    if (!bn_side_is_flat(s)) { bn_side_block_stats(s, blk, sh); return; }
    switch (s.C) {
        case 1: bn_side_block_stats_flat<1>(s, blk, sh); break;
        case 2: bn_side_block_stats_flat<2>(s, blk, sh); break;
        case 3: bn_side_block_stats_flat<3>(s, blk, sh); break;
        case 4: bn_side_block_stats_flat<4>(s, blk, sh); break;
        case 5: bn_side_block_stats_flat<5>(s, blk, sh); break;
        case 6: bn_side_block_stats_flat<6>(s, blk, sh); break;
        case 7: bn_side_block_stats_flat<7>(s, blk, sh); break;
        default: bn_side_block_stats_flat<8>(s, blk, sh); break;
    }
}

// running <- running + momentum (batch - running), unbiased variance, counter + 1 (one block)
__device__ __forceinline__ void bn_side_final(const BnSide& s, float* sh) {
    __shared__ float tot_sh[256], sq_sh[256], cnt_sh[256], mu_sh[256];
    const float* const sums = s.part;
    const float* const sqs = s.part + (size_t)s.nblk * s.C;
    const float* const cnts = s.part + (size_t)2 * s.nblk * s.C;
    bn_total(sums, s.nblk, s.C, s.CL, sh, tot_sh);
    bn_total(cnts, s.nblk, s.C, s.CL, sh, cnt_sh);
    const int col = threadIdx.x;
    const float cnt = cnt_sh[0];
    if (col < s.C) mu_sh[col] = cnt > 0.f ? tot_sh[col] / cnt : 0.f;
    __syncthreads();
    bn_total_of(s.nblk, s.C, s.CL, sh, sq_sh, [&](int b, int c) {
        const float n_b = cnts[(size_t)b * s.C + c], s_b = sums[(size_t)b * s.C + c], q_b = sqs[(size_t)b * s.C + c];     // (unconditional loads)
        const float d = s_b / fmaxf(n_b, 1.f) - mu_sh[c];
        return fmaf(n_b * d, d, q_b);
    });
    // (BatchNorm1d raises on fewer than two values per channel in training mode; a kernel cannot: such a degenerate batch -- a
    // padded batch without a single real bond -- leaves the statistics AND the counter where they are: ADVICE round 5)
    if (col < s.C && cnt > 1.f) {
        const float mu = mu_sh[col], var = sq_sh[col] / cnt;
        if (s.running_mean) s.running_mean[col] = fmaf(s.momentum, mu - s.running_mean[col], s.running_mean[col]);
        if (s.running_var) {
            const float unbiased = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
            s.running_var[col] = fmaf(s.momentum, unbiased - s.running_var[col], s.running_var[col]);
        }
    }
    if (threadIdx.x == 0 && s.nbt && cnt > 1.f) s.nbt[0] += 1;
}

// the companion alone: two launches (phase 0: block statistics, phase 2: totals), or -- one block's worth of rows -- one (phase 3)
__global__ void __launch_bounds__(256) bn_side_kernel(BnSide s, int phase) {
    __shared__ float sh[1024];
    if (phase == 0 || phase == 3) bn_side_stats_any(s, blockIdx.x, sh);
    if (phase == 3) __syncthreads();                     // (one block: its own global stores are visible to it behind a barrier)
    if (phase == 2 || phase == 3) bn_side_final(s, sh);
}

// what rides behind the statistics launch's own blocks: [touch_first, prep_first) read the hinted arrays (mkgnn_touch_hint),
// [prep_first, grid) run a pending bank preparation (mkgnn_bank_prepare_deferred) -- both independent of the batch norm, both
// otherwise paid for on the chain in front of the first convolution
template <bool VEC>
__global__ void __launch_bounds__(256) bn_stats_kernel(BnArgs a, int CL) {
    __shared__ float sh[1024];
    if (a.touch.count > 0 && (int)blockIdx.x >= a.touch_first) {     // (behind the batch norm's and the companion's own blocks)
        touch_body(a.touch, blockIdx.x - a.touch_first, gridDim.x - a.touch_first);
        return;
    }
    if (blockIdx.x >= BN_MAIN_BLOCKS) {
        bn_side_stats_any(a.side, blockIdx.x - BN_MAIN_BLOCKS, sh);
        if (a.side.nblk == 1) {                          // a companion of one block's worth of rows: all of it here, in this launch
            __syncthreads();                             // (its own global stores are visible to the block behind a barrier)
            bn_side_final(a.side, sh);
        }
        return;
    }
    bn_block_stats<VEC>(a, CL, sh);
}

template <bool VEC>
__global__ void __launch_bounds__(256) bn_stats_prep_kernel(BnArgs a, int CL, PrepManyArgs pm, int prep_first) {
    __shared__ float sh[1024];
    if ((int)blockIdx.x >= prep_first) { bank_prepare_many_block(pm, blockIdx.x - prep_first); return; }
    if ((int)blockIdx.x >= a.touch_first) {
        touch_body(a.touch, blockIdx.x - a.touch_first, prep_first - a.touch_first);
        return;
    }
    if (blockIdx.x >= BN_MAIN_BLOCKS) {
        bn_side_stats_any(a.side, blockIdx.x - BN_MAIN_BLOCKS, sh);
        if (a.side.nblk == 1) {
            __syncthreads();
            bn_side_final(a.side, sh);
        }
        return;
    }
    bn_block_stats<VEC>(a, CL, sh);
}

// Round 6, second step: ONE launch for the training forward, one for the backward -- statistics | grid-wide barrier | apply.  The
// two phases need every block's partial sums, which until now meant a kernel boundary (~5 us in a captured graph on this chip,
// for kernels that take 14 and 17).  All blocks of these launches are resident at once (at most 256 + 54 blocks of 256 threads and
// a few KB of LDS on 256 CUs), so a counter in device memory does: a block's thread 0 releases its partials (__threadfence),
// adds 1, spins until the count is the grid's size, acquires; the last block to LEAVE resets the counters.  The spin is bounded: a
// grid that is not resident within ~a second falls through (its output is then wrong, but it ends).
// MEASURED (batch 4096, profiles/r06 notes in DESIGN 4.4): 47 us for the fused forward against 14 + 17 for the two launches, the step
// 0.723 against 0.708 ms -- the agent-scope release / acquire around the counter writes back and invalidates the L2 of every XCD (the
// apply phase then re-reads from memory the rows the statistics phase had just pulled in), and 310 waves polling one counter across
// eight XCDs are slow to see it move.  Correct (the 56 batch-norm / network tests pass with it on), not faster: opt-in only,
// MKGNN_BN_ONE_LAUNCH=1.
__device__ unsigned g_bn_barrier[8];                   // {arrived, left} of the forward launch, of the backward launch
__device__ __forceinline__ void bn_grid_barrier(unsigned* ctr) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(&ctr[0], 1u);
        for (unsigned spins = 0; atomicAdd(&ctr[0], 0u) < gridDim.x && spins < (1u << 24); ++spins) __builtin_amdgcn_s_sleep(1);
        __threadfence();
        if (atomicAdd(&ctr[1], 1u) == gridDim.x - 1) { atomicExch(&ctr[0], 0u); atomicExch(&ctr[1], 0u); }      // (everybody has seen the count)
    }
    __syncthreads();
}

__device__ __forceinline__ void bn_apply_body(const BnArgs& a, int CL, float* sh) {
    __shared__ float mean[256], invstd[256], scale[256], shift[256];
    const int t = threadIdx.x;
    if (a.training) {
        const int64_t nv = bn_valid(a);
        bn_total(a.part1, bn_nblk(), a.C, CL, sh, mean);
        if (t < a.C) mean[t] = mean[t] / (float)nv;
        __syncthreads();
        bn_total_m2(a.part1, a.part2, bn_nblk(), nv, a.C, CL, mean, sh, invstd);
        if (t < a.C) {
            const float mu = mean[t], var = invstd[t] / (float)nv;
            invstd[t] = 1.f / sqrtf(var + a.eps);
            if (blockIdx.x == 0) {
                a.save_mean[t] = mu;
                a.save_invstd[t] = invstd[t];
                if (a.running_mean) a.running_mean[t] = fmaf(a.momentum, mu - a.running_mean[t], a.running_mean[t]);
                if (a.running_var) {
                    const float unbiased = nv > 1 ? var * ((float)nv / (float)(nv - 1)) : var;
                    a.running_var[t] = fmaf(a.momentum, unbiased - a.running_var[t], a.running_var[t]);
                }
            }
        }
    } else if (t < a.C) {
        mean[t] = a.running_mean[t];
        invstd[t] = 1.f / sqrtf(a.running_var[t] + a.eps);
        if (blockIdx.x == 0 && a.save_mean) { a.save_mean[t] = mean[t]; a.save_invstd[t] = invstd[t]; }
    }
    __syncthreads();
    if (t < a.C) {
        const float w = a.weight ? a.weight[t] : 1.f, b = a.bias ? a.bias[t] : 0.f;
        scale[t] = invstd[t] * w;
        shift[t] = b;
    }
    __syncthreads();
    int64_t lo, hi;
    bn_rows(a, lo, hi);
    const int c = t & (CL - 1), rsub = t / CL, RS = 256 / CL;       // same thread layout as the column sums
    if (blockIdx.x == 0 && t == 0 && a.nbt && a.training) a.nbt[0] += 1;
    if (a.inv_out) {
        // also the row norms of the output, for the kernel convolution that reads it next.  Thread layout of
        // row_inv_norm_aligned_kernel<8> (kgnn_csr.hip): eight lanes per row, four consecutive channels each (16-byte
        // loads and stores; the host checks the alignment), the same FMA chain and xor tree on the stored values --
        // bit-identical to mkgnn_row_inv_norm on `out`.
        const int l = t & 7, rs8 = t >> 3, col = 4 * l;
        const bool act = col < a.C;                       // (C is a multiple of 4 here: a lane's four channels exist or do not)
        const int cb = act ? col : 0;
        f32x4 mu4, sc4, sh4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu4[e] = mean[cb + e]; sc4[e] = scale[cb + e]; sh4[e] = shift[cb + e]; }
        for (int64_t r0 = lo + rs8; r0 - rs8 < hi; r0 += BN_UA * 32) {
            f32x4 v[BN_UA];
#pragma unroll
            for (int u = 0; u < BN_UA; ++u) v[u] = *(const f32x4*)(a.x + (r0 + u * 32 < hi ? r0 + u * 32 : hi - 1) * a.xs + cb);
#pragma unroll
            for (int u = 0; u < BN_UA; ++u) {
                const bool ok = r0 + u * 32 < hi;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = act ? fmaf(v[u][e] - mu4[e], sc4[e], sh4[e]) : 0.f;
                float ss = o[0] * o[0];
                ss = fmaf(o[1], o[1], ss); ss = fmaf(o[2], o[2], ss); ss = fmaf(o[3], o[3], ss);
                ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 1, 64);
                const float inv = 1.f / fmaxf(sqrtf(ss), MKGNN_EPS);      // (every lane of the row holds the sum)
                if (ok && l == 0) a.inv_out[r0 + u * 32] = inv;
                if (ok && act) *(f32x4*)(a.out + (r0 + u * 32) * a.os + col) = a.split_out ? split_row_store(o, inv) : o;
            }
        }
    } else if (c < a.C) {
        const float mu = mean[c], sc = scale[c], sh0 = shift[c];
        for (int64_t r0 = lo + rsub; r0 < hi; r0 += 8 * RS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.x[(r0 + u * RS < hi ? r0 + u * RS : hi - 1) * a.xs + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r0 + u * RS < hi) a.out[(r0 + u * RS) * a.os + c] = fmaf(v[u] - mu, sc, sh0);
        }
    }
}

__global__ void __launch_bounds__(256) bn_apply_kernel(BnArgs a, int CL) {
    __shared__ float sh[1024];
    if (blockIdx.x >= BN_MAIN_BLOCKS) { bn_side_final(a.side, sh); return; }      // (one extra block, training mode only)
    bn_apply_body(a, CL, sh);
}

// statistics | barrier | apply: the grid is BN_MAIN_BLOCKS + the companion's blocks (its first block does the companion's totals)
template <bool VEC>
__global__ void __launch_bounds__(256) bn_forward_fused_kernel(BnArgs a, int CL) {
    __shared__ float sh[1024];
    const bool side = blockIdx.x >= BN_MAIN_BLOCKS;
    if (side) {
        bn_side_stats_any(a.side, blockIdx.x - BN_MAIN_BLOCKS, sh);
        if (a.side.nblk == 1) { __syncthreads(); bn_side_final(a.side, sh); }
    } else {
        bn_block_stats<VEC>(a, CL, sh);
    }
    bn_grid_barrier(&g_bn_barrier[0]);
    if (side) {
        if (blockIdx.x == BN_MAIN_BLOCKS && a.side.nblk > 1) bn_side_final(a.side, sh);
        return;
    }
    bn_apply_body(a, CL, sh);
}

// backward: per-block column sums of dy and dy * xhat over the block's counted rows -> part1, part2
template <bool VEC>
__device__ __forceinline__ void bn_bwd_partial_body(const BnArgs& a, int CL, float* sh) {
    const int t = threadIdx.x;
    int64_t lo, hi;
    bn_share(bn_valid(a), bn_nblk(), blockIdx.x, lo, hi);
    float* const p1 = a.part1 + (int64_t)blockIdx.x * a.C;
    float* const p2 = a.part2 + (int64_t)blockIdx.x * a.C;
    if constexpr (VEC) {
        const int LW = CL >= 4 ? CL / 4 : 1, g = t % LW, rsub = t / LW, RS = 256 / LW, col = 4 * g;
        const bool act = col < a.C;
        const int cb = act ? col : 0;
        const f32x4 mu = {a.save_mean[cb], a.save_mean[cb + 1], a.save_mean[cb + 2], a.save_mean[cb + 3]};
        const f32x4 is = {a.save_invstd[cb], a.save_invstd[cb + 1], a.save_invstd[cb + 2], a.save_invstd[cb + 3]};
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        bn_pass_vec<2, BN_UA>(a.x, a.xs, a.gout, a.gos, lo, hi, rsub, RS, cb, mu, is, s0, s1);
        const f32x4 t0 = bn_reduce_rows(s0, LW, (f32x4*)sh);
        const f32x4 t1 = bn_reduce_rows(s1, LW, (f32x4*)sh);
        if (t < LW && act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { p1[col + e] = t0[e]; p2[col + e] = t1[e]; }
        }
    } else {
        const int c = t & (CL - 1), rsub = t / CL, RS = 256 / CL;
        const bool act = c < a.C;
        const int cc = act ? c : 0;
        float s0 = 0.f, s1 = 0.f;
        bn_pass_col<2>(a.x, a.xs, a.gout, a.gos, lo, hi, rsub, RS, cc, a.save_mean[cc], a.save_invstd[cc], s0, s1);
        const float t0 = bn_reduce_rows(s0, CL, sh);
        const float t1 = bn_reduce_rows(s1, CL, sh);
        if (t < CL && act) { p1[c] = t0; p2[c] = t1; }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(BnArgs a, int CL) {
    __shared__ float sh[1024];
    bn_bwd_partial_body<VEC>(a, CL, sh);
}

template <bool VEC>
__device__ __forceinline__ void bn_bwd_final_body(const BnArgs& a, int CL, int nblk_part, float* sh) {
    __shared__ float sdy[256], sdyx[256];
    bn_total(a.part1, nblk_part, a.C, CL, sh, sdy);
    bn_total(a.part2, nblk_part, a.C, CL, sh, sdyx);
    const int t = threadIdx.x;
    if (blockIdx.x == 0 && t < a.C) {
        if (a.gbias) a.gbias[t] = sdy[t];
        if (a.gweight) a.gweight[t] = sdyx[t];
    }
    if (!a.gx) return;
    int64_t lo, hi;
    bn_rows(a, lo, hi);
    const float invn = 1.f / (float)bn_valid(a);
    if constexpr (VEC) {
        const int LW = CL >= 4 ? CL / 4 : 1, g = t % LW, rsub = t / LW, RS = 256 / LW, col = 4 * g;
        const bool act = col < a.C;
        const int cb = act ? col : 0;
        f32x4 k0, is, mu, s0, s1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = a.weight ? a.weight[cb + e] : 1.f;
            is[e] = a.save_invstd[cb + e]; mu[e] = a.save_mean[cb + e];
            k0[e] = w * is[e]; s0[e] = sdy[cb + e] * invn; s1[e] = sdyx[cb + e] * invn;
        }
        for (int64_t r0 = lo + rsub; r0 < hi; r0 += (int64_t)BN_UA * RS) {
            f32x4 dy[BN_UA], xv[BN_UA];
#pragma unroll
            for (int u = 0; u < BN_UA; ++u) {
                const int64_t rr = r0 + (int64_t)u * RS < hi ? r0 + (int64_t)u * RS : hi - 1;
                dy[u] = *(const f32x4*)(a.gout + rr * a.gos + cb);
                xv[u] = *(const f32x4*)(a.x + rr * a.xs + cb);
            }
#pragma unroll
            for (int u = 0; u < BN_UA; ++u) {
                if (r0 + (int64_t)u * RS < hi && act) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (xv[u][e] - mu[e]) * is[e];
                        o[e] = a.training ? k0[e] * (dy[u][e] - (s0[e] + xh * s1[e])) : k0[e] * dy[u][e];
                    }
                    *(f32x4*)(a.gx + (r0 + (int64_t)u * RS) * a.gxs + col) = o;
                }
            }
        }
    } else {
        const int c = t & (CL - 1), rsub = t / CL, RS = 256 / CL;
        if (c < a.C) {
            const float w = a.weight ? a.weight[c] : 1.f;
            const float is = a.save_invstd[c], mu = a.save_mean[c];
            const float k0 = w * is, s0 = sdy[c] * invn, s1 = sdyx[c] * invn;
            for (int64_t r0 = lo + rsub; r0 < hi; r0 += 4 * RS) {
                float dy[4], xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t rr = r0 + u * RS < hi ? r0 + u * RS : hi - 1;
                    dy[u] = a.gout[rr * a.gos + c];
                    xv[u] = a.x[rr * a.xs + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (r0 + u * RS < hi) {
                        const float xh = (xv[u] - mu) * is;
                        a.gx[(r0 + u * RS) * a.gxs + c] = a.training ? k0 * (dy[u] - (s0 + xh * s1)) : k0 * dy[u];
                    }
                }
            }
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_final_kernel(BnArgs a, int CL, int nblk_part) {
    __shared__ float sh[1024];
    bn_bwd_final_body<VEC>(a, CL, nblk_part, sh);
}

// partial sums | barrier | gradient rows   (grad_x wanted: every block has rows to write)
template <bool VEC>
__global__ void __launch_bounds__(256) bn_backward_fused_kernel(BnArgs a, int CL) {
    __shared__ float sh[1024];
    bn_bwd_partial_body<VEC>(a, CL, sh);
    bn_grid_barrier(&g_bn_barrier[2]);
    bn_bwd_final_body<VEC>(a, CL, (int)gridDim.x, sh);
}

// ------------------------------------------------------------------ single-task head + BCE-with-logits ----
// The tail of the training step (reference model.py: ffn(graph_embedding) -> BCEWithLogitsLoss, mean reduction),
// ~20 tiny PyTorch kernels at B = 4096.  Two launches per pass: 32 lanes per row, 64 rows per block, per-block
// partials; a one-block kernel sums them in a fixed order.
struct HeadArgs {
    const float* emb; int64_t es; int64_t B; int H;
    const float* w; const float* b; const float* y;
    float* pred; float* loss;
    const float* gloss; float* gemb; int64_t ges; float* gw; float* gb;
    float* partial;
    float drop_p;                  // dropout on emb ahead of the product (model.py:150,169), 0 = none
    int64_t* rng;                  // forward: {seed, offset}, offset advanced by one per launch
    int64_t* rng_used;             // forward writes / backward reads the {seed, offset} of this call's mask
};
constexpr int HEAD_ROWS = 16;       // rows per block (two per half-wave: the block's latency is one row's chain, mostly its Philox rounds)

__device__ __forceinline__ float half_wave_sum(float v) {   // xor tree over the 32 lanes of a row
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256) bce_head_forward_kernel(HeadArgs a) {
    __shared__ float red[8];
    const int t = threadIdx.x, h = t & 31, g = t >> 5;          // 8 rows x 32 lanes per pass
    const float bias = a.b ? a.b[0] : 0.f;
    float s = 0.f;
    constexpr int NP = HEAD_ROWS / 8;
    // all loads of the block's 64 rows first (unconditional, clamped), then the arithmetic: one global round trip
    // per block instead of one per pass
    float xv[NP], yv[NP];
    const float w0 = h < a.H ? a.w[h] : 0.f;
    const bool drop = a.drop_p > 0.f;
    const uint64_t seed = drop ? (uint64_t)a.rng[0] : 0, offset = drop ? (uint64_t)a.rng[1] : 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
        const int64_t ic = i < a.B ? i : a.B - 1;
        xv[k] = a.emb[ic * a.es + (h < a.H ? h : 0)];
        yv[k] = a.y[ic];
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
        const int64_t ic = i < a.B ? i : a.B - 1;
        float x = h < a.H ? xv[k] * w0 : 0.f;
        if (drop) x *= keep_scale_of(seed, offset, (uint64_t)ic * a.H + (h < a.H ? h : 0), a.drop_p);
        for (int h0 = 32; h0 < a.H; h0 += 32)                   // (wider embeddings: the rare path)
            if (h0 + h < a.H) {
                float e = a.emb[ic * a.es + h0 + h];
                if (drop) e *= keep_scale_of(seed, offset, (uint64_t)ic * a.H + h0 + h, a.drop_p);
                x = fmaf(e, a.w[h0 + h], x);
            }
        x = half_wave_sum(x) + bias;
        if (h == 0 && i < a.B) {
            a.pred[i] = x;
            s += fmaxf(x, 0.f) - x * yv[k] + log1pf(expf(-fabsf(x)));   // torch's stable form
        }
    }
    if (h == 0) red[g] = s;
    __syncthreads();
    if (t == 0) {
        float p = 0.f;
        for (int k = 0; k < 8; ++k) p += red[k];
        a.partial[blockIdx.x] = p;
    }
}

// second launch of the forward: the block partials in a fixed tree -> loss; advances the dropout generator.
// (A "last block done" counter inside the first kernel did this in one launch, but the two device-scope fences it
// needs cost 10-15 us on this part -- more than a second, dependent launch: 4.7 us.)
__global__ void __launch_bounds__(256) bce_head_forward_final_kernel(HeadArgs a, int nblk) {
    __shared__ float fin[256];
    const int t = threadIdx.x;
    float v = 0.f;
    for (int bk = t; bk < nblk; bk += 256) v += a.partial[bk];
    fin[t] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) fin[t] += fin[t + w];
        __syncthreads();
    }
    if (t == 0) {
        a.loss[0] = fin[0] / (float)a.B;
        if (a.drop_p > 0.f) {
            const int64_t seed = a.rng[0], offset = a.rng[1];
            a.rng_used[0] = seed; a.rng_used[1] = offset;
            a.rng[1] = offset + 1;
        }
    }
}

__global__ void __launch_bounds__(256) bce_head_backward_kernel(HeadArgs a) {
    __shared__ float red[8][33];
    __shared__ float redb[8];
    const int t = threadIdx.x, h = t & 31, g = t >> 5;
    const float gl = a.gloss[0] / (float)a.B;
    const int PW = a.H + 1;                                   // partial row: dW[0..H), db
    float db = 0.f;
    constexpr int NP = HEAD_ROWS / 8;
    const bool drop = a.drop_p > 0.f;
    const uint64_t seed = drop ? (uint64_t)a.rng_used[0] : 0, offset = drop ? (uint64_t)a.rng_used[1] : 0;
    // d loss / d pred of the block's rows: loads first (unconditional, clamped), then the arithmetic
    float dv[NP];
    {
        float pv[NP], yv[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
            const int64_t ic = i < a.B ? i : a.B - 1;
            pv[k] = a.pred[ic];
            yv[k] = a.y[ic];
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
            dv[k] = i < a.B ? gl * (1.f / (1.f + expf(-pv[k])) - yv[k]) : 0.f;
            if (h == 0) db += dv[k];
        }
    }
    for (int h0 = 0; h0 < a.H; h0 += 32) {
        const int hh = h0 + h;
        const bool ok = hh < a.H;
        const float wv = ok ? a.w[hh] : 0.f;
        float dw = 0.f;
        float ev[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
            const int64_t ic = i < a.B ? i : a.B - 1;
            ev[k] = a.emb[ic * a.es + (ok ? hh : 0)];
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
            if (i < a.B && ok) {
                const float ks = drop ? keep_scale_of(seed, offset, (uint64_t)i * a.H + hh, a.drop_p) : 1.f;
                dw = fmaf(dv[k], ev[k] * ks, dw);
                if (a.gemb) a.gemb[i * a.ges + hh] = dv[k] * wv * ks;
            }
        }
        red[g][h] = dw;
        __syncthreads();
        if (g == 0 && ok) {
            float p = 0.f;
            for (int k = 0; k < 8; ++k) p += red[k][h];
            a.partial[(size_t)blockIdx.x * PW + hh] = p;
        }
        __syncthreads();
    }
    if (h == 0) redb[g] = db;
    __syncthreads();
    if (t == 0) {
        float p = 0.f;
        for (int k = 0; k < 8; ++k) p += redb[k];
        a.partial[(size_t)blockIdx.x * PW + a.H] = p;
    }
}

// second launch of the backward: column c of the block partials, four row parts per column, eight loads in flight
// per thread; parts combined in a fixed order
__global__ void __launch_bounds__(256) bce_head_backward_final_kernel(HeadArgs a, int nb) {
    __shared__ float fin[4][64];
    const int t = threadIdx.x;
    const int PW = a.H + 1;
    const float* part = a.partial;
    for (int c0 = 0; c0 < PW; c0 += 64) {
        const int c = c0 + (t & 63), pr = t >> 6;
        float tot = 0.f;
        if (c < PW) {
            for (int bk = pr; bk < nb; bk += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(bk + 4 * u < nb ? bk + 4 * u : bk) * PW + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (bk + 4 * u < nb) tot += v[u];
            }
        }
        fin[pr][t & 63] = tot;
        __syncthreads();
        if (pr == 0 && c < PW) {
            const float r = (fin[0][t] + fin[1][t]) + (fin[2][t] + fin[3][t]);
            if (c < a.H) a.gw[c] = r;
            else if (a.gb) a.gb[0] = r;
        }
        __syncthreads();
    }
}

// ---- forward AND the gradients for d loss = 1 in one pass (mkgnn_bce_head_fused): the loss is the end of the graph, its
// own gradient is 1 in every training step, and d loss / d pred = (sigmoid(pred) - y) / B needs nothing but the row's pred --
// so the block that computes a row's pred also writes its row of grad_emb and adds to its partials of grad_weight /
// grad_bias; ONE final kernel sums the loss and the gradient partials.  Two launches where forward + backward took four
// (the four are kept: a caller whose d loss is not 1 scales these, or runs the separate backward).
// partial row of a block: [dW[0..H) | db | loss]
__global__ void __launch_bounds__(256) bce_head_fused_kernel(HeadArgs a) {
    __shared__ float red[8][33];
    __shared__ float redb[8], redl[8];
    const int t = threadIdx.x, h = t & 31, g = t >> 5;
    const float bias = a.b ? a.b[0] : 0.f;
    const int PW = a.H + 2;
    constexpr int NP = HEAD_ROWS / 8;
    float xv[NP], yv[NP], ks0[NP], dv[NP];
    const float w0 = h < a.H ? a.w[h] : 0.f;
    const bool drop = a.drop_p > 0.f;
    const uint64_t seed = drop ? (uint64_t)a.rng[0] : 0, offset = drop ? (uint64_t)a.rng[1] : 0;
    const float invB = 1.f / (float)a.B;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
        const int64_t ic = i < a.B ? i : a.B - 1;
        xv[k] = a.emb[ic * a.es + (h < a.H ? h : 0)];
        yv[k] = a.y[ic];
    }
    float ls = 0.f, db = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
        const int64_t ic = i < a.B ? i : a.B - 1;
        ks0[k] = drop ? keep_scale_of(seed, offset, (uint64_t)ic * a.H + (h < a.H ? h : 0), a.drop_p) : 1.f;
        float x = h < a.H ? xv[k] * w0 : 0.f;
        if (drop) x *= ks0[k];
        for (int h0 = 32; h0 < a.H; h0 += 32)                   // (wider embeddings: the rare path)
            if (h0 + h < a.H) {
                float e = a.emb[ic * a.es + h0 + h];
                if (drop) e *= keep_scale_of(seed, offset, (uint64_t)ic * a.H + h0 + h, a.drop_p);
                x = fmaf(e, a.w[h0 + h], x);
            }
        x = half_wave_sum(x) + bias;                            // (the xor tree leaves the sum in every lane of the row)
        dv[k] = i < a.B ? invB * (1.f / (1.f + expf(-x)) - yv[k]) : 0.f;
        if (h == 0 && i < a.B) {
            a.pred[i] = x;
            ls += fmaxf(x, 0.f) - x * yv[k] + log1pf(expf(-fabsf(x)));   // torch's stable form
            db += dv[k];
        }
    }
    for (int h0 = 0; h0 < a.H; h0 += 32) {
        const int hh = h0 + h;
        const bool ok = hh < a.H;
        const float wv = ok ? a.w[hh] : 0.f;
        float dw = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int64_t i = (int64_t)blockIdx.x * HEAD_ROWS + k * 8 + g;
            if (i < a.B && ok) {
                float e = xv[k], ks = ks0[k];
                if (h0 > 0) {
                    e = a.emb[i * a.es + hh];
                    ks = drop ? keep_scale_of(seed, offset, (uint64_t)i * a.H + hh, a.drop_p) : 1.f;
                }
                dw = fmaf(dv[k], e * ks, dw);
                if (a.gemb) a.gemb[i * a.ges + hh] = dv[k] * wv * ks;
            }
        }
        red[g][h] = dw;
        __syncthreads();
        if (g == 0 && ok) {
            float p = 0.f;
            for (int k = 0; k < 8; ++k) p += red[k][h];
            a.partial[(size_t)blockIdx.x * PW + hh] = p;
        }
        __syncthreads();
    }
    if (h == 0) { redb[g] = db; redl[g] = ls; }
    __syncthreads();
    if (t == 0) {
        float p = 0.f, q = 0.f;
        for (int k = 0; k < 8; ++k) { p += redb[k]; q += redl[k]; }
        a.partial[(size_t)blockIdx.x * PW + a.H] = p;
        a.partial[(size_t)blockIdx.x * PW + a.H + 1] = q;
    }
}

// columns of the block partials (dW, db, loss), four row parts per column, eight loads in flight; fixed order; advances
// the dropout generator
__global__ void __launch_bounds__(256) bce_head_fused_final_kernel(HeadArgs a, int nb) {
    __shared__ float fin[4][64];
    const int t = threadIdx.x;
    const int PW = a.H + 2;
    const float* part = a.partial;
    for (int c0 = 0; c0 < PW; c0 += 64) {
        const int c = c0 + (t & 63), pr = t >> 6;
        float tot = 0.f;
        if (c < PW) {
            for (int bk = pr; bk < nb; bk += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(bk + 4 * u < nb ? bk + 4 * u : bk) * PW + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (bk + 4 * u < nb) tot += v[u];
            }
        }
        fin[pr][t & 63] = tot;
        __syncthreads();
        if (pr == 0 && c < PW) {
            const float r = (fin[0][t] + fin[1][t]) + (fin[2][t] + fin[3][t]);
            if (c < a.H) a.gw[c] = r;
            else if (c == a.H) { if (a.gb) a.gb[0] = r; }
            else a.loss[0] = r / (float)a.B;
        }
        __syncthreads();
    }
    if (t == 0 && a.drop_p > 0.f) {
        const int64_t seed = a.rng[0], offset = a.rng[1];
        a.rng_used[0] = seed; a.rng_used[1] = offset;
        a.rng[1] = offset + 1;
    }
}

}  // namespace mkgnn

using namespace mkgnn;

// ================================================================== C ABI ==========================
namespace {

struct ReadoutDims { int NT, NU, HP, FP; };

bool readout_dims(int F, int H, int G, ReadoutDims& d) {
    if (F < 1 || F > 128 || H < 1 || H > 64 || G < 1 || G > 64) return false;
    d.NT = H <= 32 ? 2 : 4;
    d.NU = F <= 64 ? 1 : 2;
    d.HP = 16 * d.NT;
    d.FP = 64 * d.NU;
    return true;
}

constexpr int RO_ATOM_BLOCKS = 256;
constexpr int RO_MOL_BLOCKS = 256;      // 16 molecules per block at batch 4096: one LDS chunk each
constexpr int BN_BLOCKS = BN_MAIN_BLOCKS;

struct ReadoutWs { size_t dA, slab_atoms, slab_mol, total; int slab_atoms_stride, slab_mol_stride; };

ReadoutWs readout_ws(const ReadoutDims& d, int H, int G, int64_t nmol) {
    ReadoutWs w;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    w.slab_atoms_stride = d.HP * d.FP + d.HP;
    w.slab_mol_stride = G * H + G + d.HP;              // (+ the db1 partial of the block-row readout)
    w.dA = 0;
    w.slab_atoms = up(w.dA + (size_t)nmol * d.HP * 4);
    w.slab_mol = up(w.slab_atoms + (size_t)2 * RO_ATOM_BLOCKS * w.slab_atoms_stride * 4);   // (block-row readout: up to 2 x)
    w.total = up(w.slab_mol + (size_t)RO_MOL_BLOCKS * w.slab_mol_stride * 4);
    return w;
}

int check_readout(const char* who, const mkgnn_readout_params* p, const float* h, int64_t h_stride, int64_t n_atoms,
                  const int32_t* mol_ptr, int64_t n_mols, ReadoutDims& d) {
    if (!p) return api_fail("%s: params is null", who);
    if (!readout_dims(p->F, p->H, p->G, d))
        return api_fail("%s: shape F=%d H=%d G=%d outside F<=128, H<=64, G<=64", who, p->F, p->H, p->G);
    if (!p->lin1_weight || !p->lin2_weight) return api_fail("%s: weight pointer is null", who);
    if (n_atoms < 0 || n_mols < 0 || n_atoms >= (int64_t)1 << 31) return api_fail("%s: bad sizes", who);
    if (h_stride < (p->F + 3) / 4 * 4 || h_stride % 4 || ((uintptr_t)h & 15))
        return api_fail("%s: h rows must be 16-byte aligned with stride >= F rounded up to 4 (stride %lld)", who,
                        (long long)h_stride);
    if (n_atoms && (!h || !mol_ptr)) return api_fail("%s: h/mol_ptr is null", who);
    return 0;
}

ReadoutArgs readout_args(const mkgnn_readout_params* p, const float* h, int64_t hs, int64_t n, const int32_t* mol_ptr,
                         const int32_t* atom_mol, int64_t nmol, const float* keep, float* pre, float* pooled) {
    ReadoutArgs a{};
    a.h = h; a.hs = hs; a.n = n; a.mol_ptr = mol_ptr; a.atom_mol = atom_mol; a.nmol = nmol;
    a.w1 = p->lin1_weight; a.b1 = p->lin1_bias; a.w2 = p->lin2_weight; a.b2 = p->lin2_bias;
    a.F = p->F; a.H = p->H; a.G = p->G;
    a.keep = keep; a.pre = pre; a.pooled = pooled;
    return a;
}

}  // namespace

extern "C" {

int32_t mkgnn_readout_hidden_stride(int32_t H) { return H <= 32 ? 32 : 64; }

size_t mkgnn_readout_workspace_bytes(int32_t F, int32_t H, int32_t G, int64_t n_atoms, int64_t n_mols) {
    ReadoutDims d;
    if (!readout_dims(F, H, G, d) || n_mols < 0) return 0;
    (void)n_atoms;
    return readout_ws(d, H, G, n_mols).total;
}

int mkgnn_readout_forward(const mkgnn_readout_params* p, const float* h, int64_t h_stride, int64_t n_atoms,
                          const int32_t* mol_ptr, int64_t n_mols, const float* keep_scale, float* pre, float* pooled,
                          float* out, int64_t out_stride, void* stream) {
    ReadoutDims d;
    if (int rc = check_readout("mkgnn_readout_forward", p, h, h_stride, n_atoms, mol_ptr, n_mols, d)) return rc;
    if (n_mols && (!out || !pooled || out_stride < p->G)) return api_fail("mkgnn_readout_forward: bad out/pooled");
    if (n_atoms && !pre) return api_fail("mkgnn_readout_forward: pre is null");
    hipStream_t st = (hipStream_t)stream;
    ReadoutArgs a = readout_args(p, h, h_stride, n_atoms, mol_ptr, nullptr, n_mols, keep_scale, pre, pooled);
    a.out = out; a.os = out_stride;
    if (n_atoms) {
        const int64_t ntiles = (n_atoms + 15) / 16;
        const int grid = (int)((ntiles + 3) / 4 < 1024 ? (ntiles + 3) / 4 : 1024);
        if (d.NT == 2 && d.NU == 1) readout_pre_kernel<2, 4><<<grid, 256, 0, st>>>(a);
        else if (d.NT == 2) readout_pre_kernel<2, 8><<<grid, 256, 0, st>>>(a);
        else if (d.NU == 1) readout_pre_kernel<4, 4><<<grid, 256, 0, st>>>(a);
        else readout_pre_kernel<4, 8><<<grid, 256, 0, st>>>(a);
    }
    if (n_mols) {
        const int grid = (int)((n_mols + 3) / 4 < 2048 ? (n_mols + 3) / 4 : 2048);
        readout_pool_kernel<<<grid, 256, 0, st>>>(a, d.HP);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_readout_forward", e);
}

int mkgnn_readout_backward(const mkgnn_readout_params* p, const float* h, int64_t h_stride, int64_t n_atoms,
                           const int32_t* mol_ptr, const int32_t* atom_mol, int64_t n_mols, const float* keep_scale,
                           const float* pre, const float* pooled, const float* grad_out, int64_t grad_out_stride,
                           float* grad_h, int64_t grad_h_stride, float* grad_lin1_weight, float* grad_lin1_bias,
                           float* grad_lin2_weight, float* grad_lin2_bias, void* ws, size_t ws_bytes, void* stream) {
    ReadoutDims d;
    if (int rc = check_readout("mkgnn_readout_backward", p, h, h_stride, n_atoms, mol_ptr, n_mols, d)) return rc;
    if (n_atoms == 0 || n_mols == 0) return api_fail("mkgnn_readout_backward: empty batch");
    if (!atom_mol || !pre || !pooled || !grad_out || grad_out_stride < p->G)
        return api_fail("mkgnn_readout_backward: null pointer or bad grad_out stride");
    if (grad_h && grad_h_stride < p->F) return api_fail("mkgnn_readout_backward: bad grad_h stride");
    const ReadoutWs w = readout_ws(d, p->H, p->G, n_mols);
    if (!ws || ws_bytes < w.total) return api_fail("mkgnn_readout_backward: workspace too small (%zu < %zu)", ws_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    ReadoutArgs a = readout_args(p, h, h_stride, n_atoms, mol_ptr, atom_mol, n_mols, keep_scale, (float*)pre, (float*)pooled);
    a.gout = grad_out; a.gos = grad_out_stride;
    a.dA = (float*)((char*)ws + w.dA);
    a.gh = grad_h; a.ghs = grad_h_stride;
    a.slab_atoms = (float*)((char*)ws + w.slab_atoms); a.slab_atoms_stride = w.slab_atoms_stride;
    a.slab_mol = (float*)((char*)ws + w.slab_mol); a.slab_mol_stride = w.slab_mol_stride;
    const int64_t ntiles = (n_atoms + 15) / 16;
    const int nw = d.NT == 4 ? 4 : 8;                // waves per block of readout_bwd_atoms_kernel
    a.nblk_atoms = (int)((ntiles + nw - 1) / nw < RO_ATOM_BLOCKS ? (ntiles + nw - 1) / nw : RO_ATOM_BLOCKS);
    a.nblk_mol = (int)((n_mols + 15) / 16 < RO_MOL_BLOCKS ? (n_mols + 15) / 16 : RO_MOL_BLOCKS);
    readout_bwd_mol_kernel<<<a.nblk_mol, 256, 0, st>>>(a, d.HP);
    if (d.NT == 2 && d.NU == 1) readout_bwd_atoms_kernel<2, 1><<<a.nblk_atoms, 512, 0, st>>>(a);
    else if (d.NT == 2) readout_bwd_atoms_kernel<2, 2><<<a.nblk_atoms, 512, 0, st>>>(a);
    else if (d.NU == 1) readout_bwd_atoms_kernel<4, 1><<<a.nblk_atoms, 256, 0, st>>>(a);
    else readout_bwd_atoms_kernel<4, 2><<<a.nblk_atoms, 256, 0, st>>>(a);
    SlabReduceArgs r{};
    int blk = 0;
    auto add = [&](const float* src, int stride, int count, int src_cols, int rows, int cols, float* dst) {
        if (!dst) return;
        SlabSeg& s = r.seg[r.nseg++];
        s.src = src; s.stride = stride; s.count = count; s.src_cols = src_cols; s.dst_rows = rows; s.dst_cols = cols;
        s.dst = dst; s.blk_start = blk;
        blk += (rows * cols + 31) / 32;
    };
    add(a.slab_atoms, a.slab_atoms_stride, a.nblk_atoms, d.FP, p->H, p->F, grad_lin1_weight);
    add(a.slab_atoms + d.HP * d.FP, a.slab_atoms_stride, a.nblk_atoms, d.HP, 1, p->H, grad_lin1_bias);
    add(a.slab_mol, a.slab_mol_stride, a.nblk_mol, p->H, p->G, p->H, grad_lin2_weight);
    add(a.slab_mol + p->G * p->H, a.slab_mol_stride, a.nblk_mol, p->G, 1, p->G, grad_lin2_bias);
    if (blk) slab_reduce_kernel<<<blk, 256, 0, st>>>(r);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_readout_backward", e);
}

// ---- block-row readout: see block_project_mfma_kernel.  Its own limits: H, G <= 64 (the pool / per-molecule kernels), K <= 255
// and every block <= 64 kernels; K may exceed the 128 columns of the dense tile-product kernels (they are not used here).
static bool blocks_dims(int K, int H, int G, ReadoutDims& d) {
    if (K < 1 || K > 255 || H < 1 || H > 64 || G < 1 || G > 64) return false;
    d.NT = H <= 32 ? 2 : 4;
    d.HP = 16 * d.NT;
    d.FP = (K + 63) / 64 * 64;
    d.NU = d.FP / 64;
    return true;
}

static int check_blocks(const char* who, const mkgnn_readout_params* p, const int32_t num_kernels[MKGNN_MAX_DEGREE],
                        const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], int64_t n_atoms, int64_t sim_stride, const float* sim,
                        BlockProjArgs& b, const ReadoutDims& d, int64_t* n_focal) {
    if (!num_kernels || !buckets) return api_fail("%s: num_kernels / buckets is null", who);
    int K = 0;
    int64_t n_bucketed = 0;
    *n_focal = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        if (num_kernels[i] < 0 || num_kernels[i] > 64) return api_fail("%s: num_kernels[%d] = %d outside 0..64", who, i, num_kernels[i]);
        b.off[i] = K; b.L[i] = num_kernels[i]; K += num_kernels[i];
        b.cnt[i] = buckets[i].count; b.sel[i] = buckets[i].selected_index;
        if (b.cnt[i] < 0 || (b.cnt[i] > 0 && !b.sel[i])) return api_fail("%s: degree %d bucket has no selected_index", who, i + 1);
        n_bucketed += b.cnt[i];
        if (num_kernels[i] > 0) *n_focal += b.cnt[i];        // (a bucket with atoms but no kernels launches no projection tile:
    }                                                        //  its z rows count as unwritten, the caller zero-fills)
    if (K != p->F) return api_fail("%s: lin1 takes %d columns, the blocks hold %d", who, p->F, K);
    if (sim_stride < K || sim_stride % 4 || !sim || ((uintptr_t)sim & 15)) return api_fail("%s: sim rows must be 16-byte aligned", who);
    if (n_bucketed > n_atoms) return api_fail("%s: the buckets hold more atoms than the batch", who);
    b.w1 = p->lin1_weight; b.H = p->H; b.K = K; b.HP = d.HP; b.FP = d.FP;
    return 0;
}

size_t mkgnn_readout_blocks_workspace_bytes(int32_t K, int32_t H, int32_t G, int64_t n_mols) {
    ReadoutDims d;
    if (!blocks_dims(K, H, G, d) || n_mols < 0) return 0;
    return readout_ws(d, H, G, n_mols).total;
}

int mkgnn_readout_blocks_supported(int32_t F, int32_t H, int32_t G, const int32_t num_kernels[MKGNN_MAX_DEGREE]) {
    ReadoutDims d;
    if (!num_kernels || !blocks_dims(F, H, G, d)) return 0;
    int K = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) { if (num_kernels[i] < 0 || num_kernels[i] > 64) return 0; K += num_kernels[i]; }
    return K == F;
}

int mkgnn_readout_blocks_forward(const mkgnn_readout_params* p, const float* sim, int64_t sim_stride,
                                 const int32_t num_kernels[MKGNN_MAX_DEGREE], const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                                 int64_t n_atoms, const int32_t* in_rowptr, const int32_t* in_col, const int32_t* mol_ptr,
                                 int64_t n_mols, const float* keep_scale, float* z, float* pre, float* pooled, float* gate_sum,
                                 float* out, int64_t out_stride, void* stream) {
    const char* who = "mkgnn_readout_blocks_forward";
    ReadoutDims d;
    if (!p || !blocks_dims(p->F, p->H, p->G, d)) return api_fail("%s: shape outside K<=255, H<=64, G<=64", who);
    if (!p->lin1_weight || !p->lin2_weight) return api_fail("%s: weight pointer is null", who);
    if (n_atoms < 1 || n_mols < 1 || n_atoms >= (int64_t)1 << 31) return api_fail("%s: bad sizes", who);
    if (!in_rowptr || !in_col || !mol_ptr || !z || !pre || !pooled || !out || out_stride < p->G)
        return api_fail("%s: null pointer or bad out stride", who);
    BlockProjArgs b{};
    int64_t n_focal = 0;
    if (int rc = check_blocks(who, p, num_kernels, buckets, n_atoms, sim_stride, sim, b, d, &n_focal)) return rc;
    hipStream_t st = (hipStream_t)stream;
    b.sim = sim; b.ss = sim_stride; b.n = n_atoms; b.z = z;
    hipError_t e = hipSuccess;
    if (n_focal < n_atoms) {                             // atoms in no bucket: their sim row is zero, and so is their z row
        e = hipMemsetAsync(z, 0, (size_t)n_atoms * d.HP * 4, st);
        if (e != hipSuccess) return api_hip_fail(who, e);
    }
    int64_t grid = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) if (b.L[i] > 0) grid += ((b.cnt[i] + 15) / 16 + 3) / 4;
    if (grid > 0) {
        const size_t lds = ((size_t)d.HP * 68 + 4 * 16 * (d.HP + 4)) * 4;
        if (d.NT == 2) block_project_mfma_kernel<2><<<(unsigned)grid, 256, lds, st>>>(b);
        else block_project_mfma_kernel<4><<<(unsigned)grid, 256, lds, st>>>(b);
    }
    // pre = propagate(z): the propagate step on H-wide rows (+ b1 where pre is read)
    e = launch_segment_sum(z, d.HP, in_rowptr, in_col, n_atoms, d.HP, pre, d.HP, nullptr, st);
    if (e != hipSuccess) return api_hip_fail(who, e);
    ReadoutArgs a = readout_args(p, nullptr, 0, n_atoms, mol_ptr, nullptr, n_mols, keep_scale, pre, pooled);
    a.out = out; a.os = out_stride; a.pre_bias = p->lin1_bias; a.gsum = gate_sum;
    const int pgrid = (int)((n_mols + 3) / 4 < 2048 ? (n_mols + 3) / 4 : 2048);
    readout_pool_kernel<<<pgrid, 256, 0, st>>>(a, d.HP);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

int mkgnn_readout_blocks_backward(const mkgnn_readout_params* p, const float* sim, int64_t sim_stride,
                                  const int32_t num_kernels[MKGNN_MAX_DEGREE], const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                                  int64_t n_atoms, const int32_t* out_rowptr, const int32_t* out_col, const int32_t* mol_ptr,
                                  const int32_t* atom_mol, int64_t n_mols, const float* gate, const float* gate_sum,
                                  const float* pooled, const float* grad_out, int64_t grad_out_stride, float* dz,
                                  float* grad_sim, int64_t grad_sim_stride, float* grad_lin1_weight, float* grad_lin1_bias,
                                  float* grad_lin2_weight, float* grad_lin2_bias, void* ws, size_t ws_bytes, void* stream) {
    const char* who = "mkgnn_readout_blocks_backward";
    ReadoutDims d;
    if (!p || !blocks_dims(p->F, p->H, p->G, d)) return api_fail("%s: shape outside K<=255, H<=64, G<=64", who);
    if (n_atoms < 1 || n_mols < 1 || n_atoms >= (int64_t)1 << 31) return api_fail("%s: bad sizes", who);
    if (!out_rowptr || !out_col || !mol_ptr || !atom_mol || !gate || !gate_sum || !pooled || !grad_out || grad_out_stride < p->G || !dz)
        return api_fail("%s: null pointer or bad grad_out stride", who);
    BlockProjArgs b{};
    int64_t n_focal = 0;
    if (int rc = check_blocks(who, p, num_kernels, buckets, n_atoms, sim_stride, sim, b, d, &n_focal)) return rc;
    if (grad_sim && grad_sim_stride < b.K) return api_fail("%s: bad grad_sim stride", who);
    const ReadoutWs w = readout_ws(d, p->H, p->G, n_mols);
    if (!ws || ws_bytes < w.total) return api_fail("%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    ReadoutArgs a = readout_args(p, nullptr, 0, n_atoms, mol_ptr, atom_mol, n_mols, nullptr, (float*)gate, (float*)pooled);
    a.gsum = (float*)gate_sum;
    a.gout = grad_out; a.gos = grad_out_stride;
    a.dA = (float*)((char*)ws + w.dA);
    a.slab_atoms = (float*)((char*)ws + w.slab_atoms); a.slab_atoms_stride = w.slab_atoms_stride;
    a.slab_mol = (float*)((char*)ws + w.slab_mol); a.slab_mol_stride = w.slab_mol_stride;
    a.nblk_mol = (int)((n_mols + 15) / 16 < RO_MOL_BLOCKS ? (n_mols + 15) / 16 : RO_MOL_BLOCKS);
    readout_bwd_mol_kernel<<<a.nblk_mol, 256, 0, st>>>(a, d.HP);
    // d loss / d z = propagate^T (d loss / d pre), d loss / d pre taken on the fly (readout_dz_gather_kernel)
    DzArgs z{};
    z.dA = a.dA; z.gate = gate; z.atom_mol = atom_mol; z.rowptr = out_rowptr; z.col = out_col; z.n = n_atoms; z.dz = dz;
    const int rpb = 256 / (d.HP / 4);
    const int nb_dz = (int)((n_atoms + rpb - 1) / rpb < DZ_BLOCKS ? (n_atoms + rpb - 1) / rpb : DZ_BLOCKS);
    if (d.NT == 2) readout_dz_gather_kernel<8><<<nb_dz, 256, 0, st>>>(z);
    else readout_dz_gather_kernel<16><<<nb_dz, 256, 0, st>>>(z);
    hipError_t e = hipSuccess;
    b.sim = sim; b.ss = sim_stride; b.n = n_atoms; b.dz = dz; b.dsim = grad_sim; b.dss = grad_sim_stride;
    // blocks bucket by bucket, 4 * tiles_per_wave tiles each; at most 2 * RO_ATOM_BLOCKS blocks (the slab capacity)
    int64_t tiles_all = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) if (b.L[i] > 0) tiles_all += (b.cnt[i] + 15) / 16;
    int tpw = (int)((tiles_all + 4 * (2 * RO_ATOM_BLOCKS - 4) - 1) / (4 * (2 * RO_ATOM_BLOCKS - 4)));
    if (tpw < 1) tpw = 1;
    int64_t nb_of[MKGNN_MAX_DEGREE], nb = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        const int64_t tiles = (b.cnt[i] + 15) / 16, per = 4 * (int64_t)tpw;
        nb_of[i] = b.L[i] > 0 ? (tiles + per - 1) / per : 0;
        nb += nb_of[i];
    }
    if (nb > 2 * RO_ATOM_BLOCKS) return api_fail("%s: internal: %lld blocks for %d slabs", who, (long long)nb, 2 * RO_ATOM_BLOCKS);
    b.slab = a.slab_atoms; b.slab_stride = a.slab_atoms_stride;
    if (nb > 0) {
        const size_t img = (size_t)4 * (16 * (d.HP + 4) + 16 * 68);
        const size_t lds = ((size_t)d.HP * 68 + (img > 4096 ? img : 4096)) * 4;
        if (d.NT == 2) block_project_bwd_mfma_kernel<2><<<(unsigned)nb, 256, lds, st>>>(b, tpw);
        else block_project_bwd_mfma_kernel<4><<<(unsigned)nb, 256, lds, st>>>(b, tpw);
    }
    SlabReduceArgs r{};
    int blk = 0;
    auto add = [&](const float* src, int stride, int count, int src_cols, int rows, int cols, float* dst, int dst_stride) {
        if (!dst || count < 1 || rows * cols < 1) return;
        SlabSeg& sg = r.seg[r.nseg++];
        sg.src = src; sg.stride = stride; sg.count = count; sg.src_cols = src_cols; sg.dst_rows = rows; sg.dst_cols = cols;
        sg.dst = dst; sg.blk_start = blk; sg.dst_stride = dst_stride;
        blk += (rows * cols + 31) / 32;
    };
    // dW1: per degree, the column window [H x L_d] of its blocks' slab images -> columns [off_d, off_d + L_d) of grad_lin1_weight
    bool absent = false;
    {
        int64_t b0 = 0;
        for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
            if (b.L[i] > 0 && nb_of[i] == 0) absent = true;
            if (nb_of[i] > 0)
                add(a.slab_atoms + (size_t)b0 * a.slab_atoms_stride + b.off[i], a.slab_atoms_stride, (int)nb_of[i], d.FP, p->H, b.L[i],
                    grad_lin1_weight ? grad_lin1_weight + b.off[i] : nullptr, b.K);
            b0 += nb_of[i];
        }
    }
    if (absent && grad_lin1_weight) {                    // a degree without atoms: its columns of the gradient are zero
        e = hipMemsetAsync(grad_lin1_weight, 0, (size_t)p->H * b.K * 4, st);
        if (e != hipSuccess) return api_hip_fail(who, e);
    }
    add(a.slab_mol + p->G * p->H + p->G, a.slab_mol_stride, a.nblk_mol, d.HP, 1, p->H, grad_lin1_bias, 0);
    add(a.slab_mol, a.slab_mol_stride, a.nblk_mol, p->H, p->G, p->H, grad_lin2_weight, 0);
    add(a.slab_mol + p->G * p->H, a.slab_mol_stride, a.nblk_mol, p->G, 1, p->G, grad_lin2_bias, 0);
    if (blk > 0) slab_reduce_kernel<<<blk, 256, 0, st>>>(r);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

// ---- the fused tail (ABI v6; kgnn_tail.hip): project | per-molecule middle | project^T + dW1 | one reduction ----
struct TailWs { size_t z, dz, slab_atoms, slab_tail, total; int slab_atoms_stride; };
static TailWs tail_ws(const ReadoutDims& d, int64_t n_atoms, int64_t n_mols) {
    TailWs w;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    w.slab_atoms_stride = d.HP * d.FP + d.HP;
    w.z = 0;
    w.dz = up(w.z + (size_t)n_atoms * 32 * 4);
    w.slab_atoms = up(w.dz + (size_t)n_atoms * 32 * 4);
    w.slab_tail = up(w.slab_atoms + (size_t)2 * RO_ATOM_BLOCKS * w.slab_atoms_stride * 4);
    w.total = up(w.slab_tail + (size_t)(n_mols < TAIL_MAX_BLOCKS ? n_mols : TAIL_MAX_BLOCKS) * TAIL_SLAB * 4);   // (an upper bound for any n_loss_mols <= n_mols)
    return w;
}

int mkgnn_tail_supported(int32_t K, int32_t H, int32_t G, const int32_t num_kernels[MKGNN_MAX_DEGREE]) {
    ReadoutDims d;
    if (!num_kernels || !blocks_dims(K, H, G, d) || H > 32 || G > 32) return 0;
    return mkgnn_readout_blocks_supported(K, H, G, num_kernels);
}

size_t mkgnn_tail_workspace_bytes(int32_t K, int32_t H, int32_t G, int64_t n_atoms, int64_t n_mols) {
    ReadoutDims d;
    if (!blocks_dims(K, H, G, d) || n_atoms < 1 || n_mols < 1) return 0;
    return tail_ws(d, n_atoms, n_mols).total;
}

int mkgnn_tail_fused(const mkgnn_tail_args* p, void* ws, size_t ws_bytes, void* stream) {
    const char* who = "mkgnn_tail_fused";
    if (!p) return api_fail("%s: null argument", who);
    const mkgnn_readout_params* ro = &p->readout;
    ReadoutDims d;
    if (!mkgnn_tail_supported(ro->F, ro->H, ro->G, p->num_kernels) || !blocks_dims(ro->F, ro->H, ro->G, d))
        return api_fail("%s: K=%d H=%d G=%d outside the fused tail (the block-row readout's shapes with H, G <= 32)", who, ro->F, ro->H, ro->G);
    if (p->n_atoms < 1 || p->n_mols < 1 || p->n_loss_mols < 1 || p->n_loss_mols > p->n_mols || p->n_atoms >= (int64_t)1 << 31)
        return api_fail("%s: bad sizes", who);
    if (!p->in_rowptr || !p->in_col || !p->out_rowptr || !p->out_col || !p->mol_ptr || !p->atom_mol || !ro->lin1_weight ||
        !ro->lin2_weight || !p->head_weight || !p->target || !p->pred || !p->loss || !p->grad_sim)
        return api_fail("%s: null pointer", who);
    if (p->dropout_p < 0.f || p->dropout_p >= 1.f) return api_fail("%s: dropout probability %g outside [0, 1)", who, (double)p->dropout_p);
    if (p->dropout_p > 0.f && (!p->rng_state || !p->rng_used)) return api_fail("%s: dropout needs rng_state and rng_used", who);
    if (p->emb && p->emb_stride < ro->G) return api_fail("%s: bad emb stride", who);
    BlockProjArgs b{};
    int64_t n_focal = 0;
    if (int rc = check_blocks(who, ro, p->num_kernels, p->buckets, p->n_atoms, p->sim_stride, p->sim, b, d, &n_focal)) return rc;
    if (p->grad_sim_stride < b.K) return api_fail("%s: bad grad_sim stride", who);
    const TailWs w = tail_ws(d, p->n_atoms, p->n_mols);
    if (!ws || ws_bytes < w.total) return api_fail("%s: workspace too small (%zu < %zu)", who, ws_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    float* const z = (float*)((char*)ws + w.z);
    float* const dz = (float*)((char*)ws + w.dz);
    hipError_t e = hipSuccess;
    // (1) z = W1[:, block] sim[block]   (d.HP = 32: H <= 32)
    b.sim = p->sim; b.ss = p->sim_stride; b.n = p->n_atoms; b.z = z;
    if (n_focal < p->n_atoms) {                          // atoms in no bucket: their sim row is zero, and so is their z row
        e = hipMemsetAsync(z, 0, (size_t)p->n_atoms * d.HP * 4, st);
        if (e != hipSuccess) return api_hip_fail(who, e);
    }
    int64_t grid = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) if (b.L[i] > 0) grid += ((b.cnt[i] + 15) / 16 + 3) / 4;
    if (grid > 0) {
        const size_t lds = ((size_t)d.HP * 68 + 4 * 16 * (d.HP + 4)) * 4;
        block_project_mfma_kernel<2><<<(unsigned)grid, 256, lds, st>>>(b);
    }
    // (2) the middle: propagate, swish, pool, lin2, head, loss and the way back to d z, per chunk of whole molecules
    TailMidArgs m{};
    m.z = z; m.dz = dz; m.rin = p->in_rowptr; m.cin = p->in_col; m.rout = p->out_rowptr; m.cout = p->out_col;
    m.mol_ptr = p->mol_ptr; m.atom_mol = p->atom_mol; m.n_atoms = p->n_atoms; m.n_mols = p->n_mols; m.n_loss = p->n_loss_mols;
    m.b1 = ro->lin1_bias; m.w2 = ro->lin2_weight; m.b2 = ro->lin2_bias; m.wh = p->head_weight; m.bh = p->head_bias; m.y = p->target;
    m.H = ro->H; m.G = ro->G; m.drop_p = p->dropout_p; m.rng = p->rng_state;
    m.emb = p->emb; m.es = p->emb_stride; m.pred = p->pred;
    m.slab = (float*)((char*)ws + w.slab_tail); m.slab_stride = TAIL_SLAB;
    m.mg = tail_group_size(p->n_loss_mols);
    const int nbm = tail_middle_blocks(p->n_loss_mols);
    e = launch_tail_middle(m, nbm, st);
    if (e != hipSuccess) return api_hip_fail(who, e);
    // (3) d sim[block] = W1[:, block]^T d z,  dW1 partials   (as mkgnn_readout_blocks_backward)
    b.dz = dz; b.dsim = p->grad_sim; b.dss = p->grad_sim_stride;
    int64_t tiles_all = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) if (b.L[i] > 0) tiles_all += (b.cnt[i] + 15) / 16;
    int tpw = (int)((tiles_all + 4 * (2 * RO_ATOM_BLOCKS - 4) - 1) / (4 * (2 * RO_ATOM_BLOCKS - 4)));
    if (tpw < 1) tpw = 1;
    int64_t nb_of[MKGNN_MAX_DEGREE], nb = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        const int64_t tiles = (b.cnt[i] + 15) / 16, per = 4 * (int64_t)tpw;
        nb_of[i] = b.L[i] > 0 ? (tiles + per - 1) / per : 0;
        nb += nb_of[i];
    }
    if (nb > 2 * RO_ATOM_BLOCKS) return api_fail("%s: internal: %lld blocks for %d slabs", who, (long long)nb, 2 * RO_ATOM_BLOCKS);
    float* const slab_atoms = (float*)((char*)ws + w.slab_atoms);
    b.slab = slab_atoms; b.slab_stride = w.slab_atoms_stride;
    if (nb > 0) {
        const size_t img = (size_t)4 * (16 * (d.HP + 4) + 16 * 68);
        const size_t lds = ((size_t)d.HP * 68 + (img > 4096 ? img : 4096)) * 4;
        block_project_bwd_mfma_kernel<2><<<(unsigned)nb, 256, lds, st>>>(b, tpw);
    }
    // (4) every partial slab -> its gradient, the loss; the dropout generator advances
    SlabReduceArgs r{};
    int blk = 0;
    auto add = [&](const float* src, int stride, int count, int src_cols, int rows, int cols, float* dst, int dst_stride) {
        if (!dst || count < 1 || rows * cols < 1) return;
        SlabSeg& sg = r.seg[r.nseg++];
        sg.src = src; sg.stride = stride; sg.count = count; sg.src_cols = src_cols; sg.dst_rows = rows; sg.dst_cols = cols;
        sg.dst = dst; sg.blk_start = blk; sg.dst_stride = dst_stride;
        blk += (rows * cols + 31) / 32;
    };
    bool absent = false;
    {
        int64_t b0 = 0;
        for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
            if (b.L[i] > 0 && nb_of[i] == 0) absent = true;
            if (nb_of[i] > 0)
                add(slab_atoms + (size_t)b0 * w.slab_atoms_stride + b.off[i], w.slab_atoms_stride, (int)nb_of[i], d.FP, ro->H, b.L[i],
                    p->grad_lin1_weight ? p->grad_lin1_weight + b.off[i] : nullptr, b.K);
            b0 += nb_of[i];
        }
    }
    if (absent && p->grad_lin1_weight) {                 // a degree without atoms: its columns of the gradient are zero
        e = hipMemsetAsync(p->grad_lin1_weight, 0, (size_t)ro->H * b.K * 4, st);
        if (e != hipSuccess) return api_hip_fail(who, e);
    }
    add(m.slab + TAIL_B1, TAIL_SLAB, nbm, 32, 1, ro->H, p->grad_lin1_bias, 0);
    add(m.slab + TAIL_W2, TAIL_SLAB, nbm, 32, ro->G, ro->H, p->grad_lin2_weight, 0);
    add(m.slab + TAIL_B2, TAIL_SLAB, nbm, 32, 1, ro->G, p->grad_lin2_bias, 0);
    add(m.slab + TAIL_WH, TAIL_SLAB, nbm, 32, 1, ro->G, p->grad_head_weight, 0);
    add(m.slab + TAIL_BH, TAIL_SLAB, nbm, 1, 1, 1, p->grad_head_bias, 0);
    add(m.slab + TAIL_LOSS, TAIL_SLAB, nbm, 1, 1, 1, p->loss, 0);
    r.drop_p = p->dropout_p; r.rng = p->rng_state; r.rng_used = p->rng_used;
    // (a reduction an earlier call left pending and nobody took: now, in front of this one)
    e = launch_pending_tail_reduce(st);
    if (e != hipSuccess) return api_hip_fail(who, e);
    if (p->defer_reduce && blk > 0) {                    // round 6: off the critical chain -- see mkgnn_tail_args.defer_reduce
        std::lock_guard<std::mutex> lock(g_pending_reduce_mutex);
        PendingReduce* slot = pending_reduce_slot();
        slot->r = r; slot->blk = blk;
        return 0;
    }
    if (blk > 0) slab_reduce_kernel<<<blk, 256, 0, st>>>(r);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

extern "C++" {
namespace mkgnn {
hipError_t launch_pending_tail_reduce(hipStream_t st, bool* launched) {
    if (launched) *launched = false;
    std::lock_guard<std::mutex> lock(g_pending_reduce_mutex);
    PendingReduce* slot = pending_reduce_slot();
    if (slot->blk <= 0) return hipSuccess;
    const int blk = slot->blk;
    slot->blk = 0;
    slab_reduce_kernel<<<blk, 256, 0, st>>>(slot->r);
    if (launched) *launched = true;
    return hipGetLastError();
}
}  // namespace mkgnn
}

extern "C" int mkgnn_tail_flush(void* stream) {
    const hipError_t e = launch_pending_tail_reduce((hipStream_t)stream);
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_tail_flush", e);
}

size_t mkgnn_batchnorm_workspace_bytes(int32_t C) { return C > 0 ? (size_t)2 * BN_BLOCKS * C * 4 : 0; }

static int bn_common(const char* who, int64_t n, int32_t C, int& CL) {
    if (C < 1 || C > 256) return api_fail("%s: channel count C=%d outside 1..256", who, C);
    if (n < 1) return api_fail("%s: batch norm needs at least one row", who);
    CL = 1;
    while (CL < C) CL <<= 1;
    return 0;
}

size_t mkgnn_batchnorm_stats_workspace_bytes(int32_t C) { return C > 0 ? (size_t)3 * BN_SIDE_BLOCKS * C * 4 : 0; }

// checks a companion and turns it into the kernels' form; s.nblk blocks
static int bn_side_setup(const char* who, const mkgnn_bn_stats* c, void* ws, size_t ws_bytes, BnSide& s) {
    if (!c->x || c->n_rows < 1 || c->C < 1 || c->C > 256 || c->x_stride < c->C) return api_fail("%s: bad companion tensor", who);
    if ((c->row_key != nullptr) != (c->key_limit != nullptr)) return api_fail("%s: row_key and key_limit come together", who);
    if (!ws || ws_bytes < mkgnn_batchnorm_stats_workspace_bytes(c->C)) return api_fail("%s: companion workspace too small", who);
    s.x = c->x; s.xs = c->x_stride; s.n = c->n_rows; s.C = c->C;
    s.CL = 1;
    while (s.CL < s.C) s.CL <<= 1;
    s.running_mean = c->running_mean; s.running_var = c->running_var; s.momentum = c->momentum; s.nbt = c->num_batches_tracked;
    s.key = c->row_key; s.key_limit = c->key_limit; s.part = (float*)ws;
    s.flat = (c->C <= 8 && c->x_stride == c->C && ((uintptr_t)c->x & 15) == 0) ? 1 : 0;
    const int64_t rows_per_pass = s.flat ? SIDE_FLAT_ROWS : 8 * (256 / s.CL);      // rows a block takes per loop trip
    int64_t nb = (c->n_rows + rows_per_pass - 1) / rows_per_pass;
    s.nblk = (int)(nb < 1 ? 1 : (nb > BN_SIDE_BLOCKS ? BN_SIDE_BLOCKS : nb));
    return 0;
}

int mkgnn_batchnorm_update_stats(const mkgnn_bn_stats* c, void* ws, size_t ws_bytes, void* stream) {
    const char* who = "mkgnn_batchnorm_update_stats";
    if (!c) return api_fail("%s: null argument", who);
    BnSide s{};
    if (int rc = bn_side_setup(who, c, ws, ws_bytes, s)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)c->n_rows * s.CL <= 64 * 1024) {         // one block does all three steps in one launch
        s.nblk = 1;
        bn_side_kernel<<<1, 256, 0, st>>>(s, 3);
    } else {
        bn_side_kernel<<<s.nblk, 256, 0, st>>>(s, 0);
        bn_side_kernel<<<1, 256, 0, st>>>(s, 2);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

int mkgnn_batchnorm_forward(const float* x, int64_t x_stride, int64_t n_rows, int32_t C, const float* weight,
                            const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                            int32_t training, float* out, int64_t out_stride, float* save_mean, float* save_invstd,
                            float* inv_norm, int64_t* num_batches_tracked, const int64_t* n_valid_rows, void* ws, size_t ws_bytes,
                            void* stream) {
    return mkgnn_batchnorm_forward_with_stats(x, x_stride, n_rows, C, weight, bias, running_mean, running_var, momentum, eps, training,
                                              out, out_stride, save_mean, save_invstd, inv_norm, num_batches_tracked, n_valid_rows,
                                              ws, ws_bytes, nullptr, nullptr, 0, stream);
}

int mkgnn_batchnorm_forward_with_stats(const float* x, int64_t x_stride, int64_t n_rows, int32_t C, const float* weight,
                            const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                            int32_t training, float* out, int64_t out_stride, float* save_mean, float* save_invstd,
                            float* inv_norm, int64_t* num_batches_tracked, const int64_t* n_valid_rows, void* ws, size_t ws_bytes,
                            const mkgnn_bn_stats* companion, void* companion_ws, size_t companion_ws_bytes, void* stream) {
    int CL;
    if (int rc = bn_common("mkgnn_batchnorm_forward", n_rows, C, CL)) return rc;
    const bool split_rows = (training & MKGNN_BN_SPLIT_ROWS) != 0;       // out is written pre-split (kgnn_split.h)
    training &= 1;
    if (split_rows && !inv_norm) return api_fail("mkgnn_batchnorm_forward: MKGNN_BN_SPLIT_ROWS needs inv_norm");
    if (companion && !training) return api_fail("mkgnn_batchnorm_forward: a statistics companion needs training mode (nothing moves in eval mode)");
    if (!x || !out || x_stride < C || out_stride < C) return api_fail("mkgnn_batchnorm_forward: bad x/out");
    if (training && (!save_mean || !save_invstd)) return api_fail("mkgnn_batchnorm_forward: save_mean/save_invstd is null");
    if (!training && (!running_mean || !running_var)) return api_fail("mkgnn_batchnorm_forward: eval mode needs running statistics");
    if (training && (!ws || ws_bytes < mkgnn_batchnorm_workspace_bytes(C))) return api_fail("mkgnn_batchnorm_forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    BnArgs a{};
    a.x = x; a.xs = x_stride; a.n = n_rows; a.C = C; a.weight = weight; a.bias = bias;
    a.running_mean = running_mean; a.running_var = running_var; a.momentum = momentum; a.eps = eps; a.training = training;
    a.out = out; a.os = out_stride; a.save_mean = save_mean; a.save_invstd = save_invstd;
    if (inv_norm && (C > 32 || C % 4 || x_stride % 4 || out_stride % 4 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)))
        return api_fail("mkgnn_batchnorm_forward: inv_norm needs C <= 32, a multiple of 4, and 16-byte aligned rows of x and out (C=%d)", C);
    a.inv_out = inv_norm; a.split_out = split_rows ? 1 : 0; a.nbt = num_batches_tracked; a.nvalid = n_valid_rows;
    a.part1 = (float*)ws; a.part2 = a.part1 ? a.part1 + (size_t)BN_BLOCKS * C : nullptr;
    static_assert(BN_BLOCKS == BN_MAIN_BLOCKS, "the companion's blocks sit behind the batch norm's own");
    bool side_single = false;
    if (companion) {
        if (int rc = bn_side_setup("mkgnn_batchnorm_forward", companion, companion_ws, companion_ws_bytes, a.side)) return rc;
        if ((int64_t)companion->n_rows * a.side.CL <= 64 * 1024) { a.side.nblk = 1; side_single = true; }
    }
    static const bool one_launch = [] { const char* e = getenv("MKGNN_BN_ONE_LAUNCH"); return e && e[0] == '1'; }();      // (opt-in: it lost)
    if (training && one_launch) {                        // statistics | grid barrier | apply  (MKGNN_BN_ONE_LAUNCH=0: two launches)
        if (bn_vec_rows(x, x_stride, C)) bn_forward_fused_kernel<true><<<BN_BLOCKS + a.side.nblk, 256, 0, st>>>(a, CL);
        else bn_forward_fused_kernel<false><<<BN_BLOCKS + a.side.nblk, 256, 0, st>>>(a, CL);
    } else {
        if (training) {                                  // block statistics (the companion's blocks behind the batch norm's own)
            // ... and behind those the blocks of a pending mkgnn_touch_hint: this launch is bound by the latency of its two passes
            // over 11 MB, not by bandwidth -- the batch's index arrays are read beside them for nothing
            a.touch.count = 0;
            a.touch_first = BN_BLOCKS + a.side.nblk;
            const int extra = take_touch_hint(a.touch) ? TOUCH_BLOCKS : 0;
            PrepManyArgs pm;
            if (take_pending_prepare(pm)) {              // ... and behind those a pending bank preparation's tasks
                pm.touch.count = 0;
                const int prep_first = a.touch_first + extra;
                if (bn_vec_rows(x, x_stride, C)) bn_stats_prep_kernel<true><<<prep_first + pm.prep_blocks, 256, 0, st>>>(a, CL, pm, prep_first);
                else bn_stats_prep_kernel<false><<<prep_first + pm.prep_blocks, 256, 0, st>>>(a, CL, pm, prep_first);
            } else if (bn_vec_rows(x, x_stride, C)) bn_stats_kernel<true><<<BN_BLOCKS + a.side.nblk + extra, 256, 0, st>>>(a, CL);
            else bn_stats_kernel<false><<<BN_BLOCKS + a.side.nblk + extra, 256, 0, st>>>(a, CL);
            a.touch.count = 0;
        }
        bn_apply_kernel<<<BN_BLOCKS + ((a.side.nblk && !side_single) ? 1 : 0), 256, 0, st>>>(a, CL);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_batchnorm_forward", e);
}

int mkgnn_batchnorm_backward(const float* grad_out, int64_t grad_out_stride, const float* x, int64_t x_stride,
                             int64_t n_rows, int32_t C, const float* weight, const float* save_mean,
                             const float* save_invstd, int32_t training, float* grad_x, int64_t grad_x_stride, float* grad_weight, float* grad_bias,
                             const int64_t* n_valid_rows, void* ws, size_t ws_bytes, void* stream) {
    int CL;
    if (int rc = bn_common("mkgnn_batchnorm_backward", n_rows, C, CL)) return rc;
    if (!grad_out || !x || grad_out_stride < C || x_stride < C) return api_fail("mkgnn_batchnorm_backward: bad grad_out/x");
    if (!save_mean || !save_invstd) return api_fail("mkgnn_batchnorm_backward: saved statistics are null");
    if (grad_x && grad_x_stride < C) return api_fail("mkgnn_batchnorm_backward: bad grad_x stride");
    if (!ws || ws_bytes < mkgnn_batchnorm_workspace_bytes(C)) return api_fail("mkgnn_batchnorm_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    BnArgs a{};
    a.x = x; a.xs = x_stride; a.n = n_rows; a.C = C; a.weight = weight;
    a.training = training;
    a.save_mean = (float*)save_mean; a.save_invstd = (float*)save_invstd;
    a.part1 = (float*)ws; a.part2 = a.part1 + (size_t)BN_BLOCKS * C;
    a.gout = grad_out; a.gos = grad_out_stride; a.gx = grad_x; a.gxs = grad_x_stride;
    a.gweight = grad_weight; a.gbias = grad_bias; a.nvalid = n_valid_rows;
    static const bool one_launch = [] { const char* e = getenv("MKGNN_BN_ONE_LAUNCH"); return e && e[0] == '1'; }();      // (opt-in: it lost)
    const bool vec = bn_vec_rows(x, x_stride, C) && bn_vec_rows(grad_out, grad_out_stride, C) && (!grad_x || bn_vec_rows(grad_x, grad_x_stride, C));
    if (one_launch && grad_x) {                          // partial sums | grid barrier | gradient rows
        if (vec) bn_backward_fused_kernel<true><<<BN_BLOCKS, 256, 0, st>>>(a, CL);
        else bn_backward_fused_kernel<false><<<BN_BLOCKS, 256, 0, st>>>(a, CL);
    } else if (vec) {
        bn_bwd_partial_kernel<true><<<BN_BLOCKS, 256, 0, st>>>(a, CL);
        bn_bwd_final_kernel<true><<<grad_x ? BN_BLOCKS : 1, 256, 0, st>>>(a, CL, BN_BLOCKS);
    } else {
        bn_bwd_partial_kernel<false><<<BN_BLOCKS, 256, 0, st>>>(a, CL);
        bn_bwd_final_kernel<false><<<grad_x ? BN_BLOCKS : 1, 256, 0, st>>>(a, CL, BN_BLOCKS);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail("mkgnn_batchnorm_backward", e);
}

size_t mkgnn_bce_head_workspace_bytes(int64_t n_rows, int32_t H) {
    if (n_rows < 1 || H < 1) return 0;
    return 16 + (size_t)((n_rows + HEAD_ROWS - 1) / HEAD_ROWS) * (H + 2) * 4;
}

int mkgnn_bce_head_fused(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight, const float* bias,
                         const float* target, float dropout_p, int64_t* rng_state, int64_t* rng_used, float* pred, float* loss,
                         float* grad_emb, int64_t grad_emb_stride, float* grad_weight, float* grad_bias, void* ws,
                         size_t ws_bytes, void* stream) {
    const char* who = "mkgnn_bce_head_fused";
    if (n_rows < 1 || H < 1 || emb_stride < H) return api_fail("%s: bad shape", who);
    if (!emb || !weight || !target || !pred || !loss || !grad_weight) return api_fail("%s: null pointer", who);
    if (grad_emb && grad_emb_stride < H) return api_fail("%s: bad grad_emb stride", who);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return api_fail("%s: dropout probability %g outside [0, 1)", who, dropout_p);
    if (dropout_p > 0.f && (!rng_state || !rng_used)) return api_fail("%s: dropout needs rng_state and rng_used", who);
    HeadArgs a{};
    if (!ws || ws_bytes < mkgnn_bce_head_workspace_bytes(n_rows, H) || ((uintptr_t)ws & 3))
        return api_fail("%s: workspace too small or misaligned", who);
    a.partial = (float*)((char*)ws + 16);
    a.emb = emb; a.es = emb_stride; a.B = n_rows; a.H = H; a.w = weight; a.b = bias; a.y = target; a.pred = pred; a.loss = loss;
    a.gemb = grad_emb; a.ges = grad_emb_stride; a.gw = grad_weight; a.gb = grad_bias;
    a.drop_p = dropout_p; a.rng = rng_state; a.rng_used = rng_used;
    const int nblk = (int)((n_rows + HEAD_ROWS - 1) / HEAD_ROWS);
    bce_head_fused_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(a);
    bce_head_fused_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, nblk);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

static int head_ws(const char* who, int64_t n_rows, int32_t H, void* ws, size_t ws_bytes, HeadArgs& a) {
    if (!ws || ws_bytes < mkgnn_bce_head_workspace_bytes(n_rows, H) || ((uintptr_t)ws & 3))
        return api_fail("%s: workspace too small or misaligned", who);
    a.partial = (float*)((char*)ws + 16);
    return 0;
}

static int head_forward(const char* who, const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight,
                        const float* bias, const float* target, float p, int64_t* rng_state, int64_t* rng_used, float* pred,
                        float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (n_rows < 1 || H < 1 || emb_stride < H) return api_fail("%s: bad shape", who);
    if (!emb || !weight || !target || !pred || !loss) return api_fail("%s: null pointer", who);
    if (!(p >= 0.f && p < 1.f)) return api_fail("%s: dropout probability %g outside [0, 1)", who, p);
    if (p > 0.f && (!rng_state || !rng_used)) return api_fail("%s: dropout needs rng_state and rng_used", who);
    HeadArgs a{};
    if (int rc = head_ws(who, n_rows, H, ws, ws_bytes, a)) return rc;
    a.emb = emb; a.es = emb_stride; a.B = n_rows; a.H = H; a.w = weight; a.b = bias; a.y = target; a.pred = pred; a.loss = loss;
    a.drop_p = p; a.rng = rng_state; a.rng_used = rng_used;
    const int nblk = (int)((n_rows + HEAD_ROWS - 1) / HEAD_ROWS);
    bce_head_forward_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(a);
    bce_head_forward_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, nblk);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

static int head_backward(const char* who, const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight,
                         const float* target, const float* pred, const float* grad_loss, float p, const int64_t* rng_used,
                         float* grad_emb, int64_t grad_emb_stride, float* grad_weight, float* grad_bias, void* ws,
                         size_t ws_bytes, void* stream) {
    if (n_rows < 1 || H < 1 || emb_stride < H) return api_fail("%s: bad shape", who);
    if (!emb || !weight || !target || !pred || !grad_loss || !grad_weight) return api_fail("%s: null pointer", who);
    if (grad_emb && grad_emb_stride < H) return api_fail("%s: bad grad_emb stride", who);
    if (!(p >= 0.f && p < 1.f)) return api_fail("%s: dropout probability %g outside [0, 1)", who, p);
    if (p > 0.f && !rng_used) return api_fail("%s: dropout needs the forward's rng_used", who);
    HeadArgs a{};
    if (int rc = head_ws(who, n_rows, H, ws, ws_bytes, a)) return rc;
    a.emb = emb; a.es = emb_stride; a.B = n_rows; a.H = H; a.w = weight; a.y = target; a.pred = (float*)pred;
    a.gloss = grad_loss; a.gemb = grad_emb; a.ges = grad_emb_stride; a.gw = grad_weight; a.gb = grad_bias;
    a.drop_p = p; a.rng_used = (int64_t*)rng_used;
    const int nblk = (int)((n_rows + HEAD_ROWS - 1) / HEAD_ROWS);
    bce_head_backward_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(a);
    bce_head_backward_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, nblk);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

int mkgnn_bce_head_forward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight,
                           const float* bias, const float* target, float* pred, float* loss, void* ws, size_t ws_bytes,
                           void* stream) {
    return head_forward("mkgnn_bce_head_forward", emb, emb_stride, n_rows, H, weight, bias, target, 0.f, nullptr, nullptr,
                        pred, loss, ws, ws_bytes, stream);
}

int mkgnn_bce_head_backward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight,
                            const float* target, const float* pred, const float* grad_loss, float* grad_emb,
                            int64_t grad_emb_stride, float* grad_weight, float* grad_bias, void* ws, size_t ws_bytes,
                            void* stream) {
    return head_backward("mkgnn_bce_head_backward", emb, emb_stride, n_rows, H, weight, target, pred, grad_loss, 0.f, nullptr,
                         grad_emb, grad_emb_stride, grad_weight, grad_bias, ws, ws_bytes, stream);
}

int mkgnn_bce_head_dropout_forward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight,
                                   const float* bias, const float* target, float dropout_p, int64_t* rng_state,
                                   int64_t* rng_used, float* pred, float* loss, void* ws, size_t ws_bytes, void* stream) {
    return head_forward("mkgnn_bce_head_dropout_forward", emb, emb_stride, n_rows, H, weight, bias, target, dropout_p, rng_state,
                        rng_used, pred, loss, ws, ws_bytes, stream);
}

int mkgnn_bce_head_dropout_backward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H, const float* weight,
                                    const float* target, const float* pred, const float* grad_loss, float dropout_p,
                                    const int64_t* rng_used, float* grad_emb, int64_t grad_emb_stride, float* grad_weight,
                                    float* grad_bias, void* ws, size_t ws_bytes, void* stream) {
    return head_backward("mkgnn_bce_head_dropout_backward", emb, emb_stride, n_rows, H, weight, target, pred, grad_loss, dropout_p,
                         rng_used, grad_emb, grad_emb_stride, grad_weight, grad_bias, ws, ws_bytes, stream);
}

}  // extern "C"
