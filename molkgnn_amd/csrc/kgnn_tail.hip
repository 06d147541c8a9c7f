// The tail of a training step in ONE launch (round 6): everything behind the last kernel convolution --
//
//   h = propagate(sim_sc)                                   reference KernelLayer.py:119-123
//   emb_g = pool_g( lin2( swish( lin1(h) ) ) )              reference MolKGNNNet.py:144-146
//   loss = BCEWithLogits( ffn( dropout(emb) ), y )          reference model.py:147-150, 169, 190-198; data.py:37
//
// -- forward AND backward: the loss is a mean over molecules, so d loss / d logit_g = (sigmoid(logit_g) - y_g) / B needs nothing
// but the molecule's own logit, and the whole chain back to d loss / d sim_sc (block rows, what the last convolution's backward
// reads) and the six parameter gradients can be taken while the molecule is still in LDS.  Rounds 3-5 ran this as NINE launches
// (project, propagate, pool, head, head reduce | molecule gradients, propagate^T, project^T + dW1, slab reduce: 102 us of a 725 us
// step at batch 4096, about half of it launch latency -- a dependent kernel costs ~5 us in a captured graph on this chip whatever
// it does); the same arithmetic here is one launch plus the fixed-order reduction of the per-block gradient slabs.
//
// A workgroup takes a contiguous run of whole molecules and walks it in CHUNKS of molecules that fit LDS (<= AC atoms, <= EC
// edges each way, <= MC molecules): bulk loads of the chunk's indices and of every atom's OWN column block of sim (a row of a
// kernel convolution's output is non-zero only there: 10 / 20 / 30 / 50 of 110 floats), then
//     z = W1[:, block] sim[block]            per atom                (lin1 commutes with the neighbour sum: project first)
//     pre = b1 + sum of z over the in-edges  per atom, from LDS      (propagate on H-wide rows)
//     e_g = sum_n swish(pre_n)               per molecule            (lin2 commutes with the pool: one row per molecule)
//     emb_g = W2 e_g + |g| b2,  logit, loss, d logit;  d emb, d e_g  per molecule
//     d pre = d e_g * swish'(pre);  d z = sum of d pre over the out-edges;  d sim[block] = W1[:, block]^T d z;  dW1 += d z (x) sim
// with block-wide barriers between the phases.  Molecules never share atoms or edges, so a chunk needs nothing from outside.
// No float atomics: every sum runs in a fixed order, the per-workgroup partial gradients go to slabs that a second small
// kernel adds up in block order.  The dropout mask of the head comes from the same counter-based generator, element for
// element, as mkgnn_bce_head_fused (kgnn_philox.h).
//
// Limits (mkgnn_tail_supported; the host falls back to the nine launches otherwise): K <= 112 columns, every degree's block
// <= 52, H <= 32, G <= 32, no dropout inside the readout, and -- checked per batch by the host, which knows the molecule
// sizes -- no single molecule beyond a chunk (128 atoms, 512 edges).  A molecule that breaks the last promise is skipped and
// the loss comes back NaN: loud, not wrong.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <math.h>

#include "kgnn_common.h"
#include "kgnn_launch.h"
#include "kgnn_philox.h"
#include "../../include/molkgnn_hip.h"

namespace mkgnn {
namespace tail {

constexpr int NT = 256;                 // threads per workgroup: 8 row slots x 32 hidden lanes
constexpr int AC = MKGNN_TAIL_MAX_ATOMS, EC = MKGNN_TAIL_MAX_EDGES, MC = 32;      // chunk capacity: atoms, edges (each way), molecules
constexpr int LB = 52, LBP = 53;        // widest column block; its LDS pitch (odd: a column walk is conflict-free)
constexpr int KMAX = 112, HP = 33;      // columns of sim; pitch of the 32-wide rows
constexpr int CG = 14;                  // dW1 columns per thread: 8 slots x 14 = 112

struct Args {
    const float* sim; int64_t ss;
    const int8_t* deg;
    const int32_t *rin, *cin, *rout, *cout;
    const int32_t *mol_ptr, *atom_mol;
    int64_t n_atoms, n_mols, n_loss;
    const float *w1, *b1, *w2, *b2, *wh, *bh, *y;
    int K, H, G;
    uint32_t blk_off, blk_len;          // byte d: first column / width of degree d's block (d = 1..4 -> bytes 0..3)
    float drop_p; const int64_t* rng;
    float* emb; int64_t es;             // [n_mols, G] or null
    float* pred; float* gsim; int64_t gs;
    float* slab; int slab_stride;       // [gridDim.x][slab_stride]: W1 [H][K] | b1 [32] | W2 [32][32] | b2 [32] | wh [32] | bh | loss
    int* status;                        // != 0 after the launch: a molecule did not fit a chunk
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ float half_sum(float v) {       // xor tree over the 32 lanes of a row slot
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float lane_of(float v, int k) { return __shfl(v, (int)(threadIdx.x & 32) | k, 64); }    // lane k of my row slot

__host__ __device__ constexpr int lds_floats() {
    return KMAX * HP + 32 * HP + AC * LBP + 2 * AC * HP + MC * HP + 96;
}
__host__ __device__ constexpr int lds_ints() { return 2 * AC + 2 * (AC + 1) + 2 * EC + 3 * (MC + 1) + 4; }

__global__ void __launch_bounds__(NT) tail_fused_kernel(Args a) {
    extern __shared__ __align__(16) float lds[];
    float* const W1t = lds;                              // [K][HP]: W1t[c][j] = W1[j][c]
    float* const W2s = W1t + KMAX * HP;                  // [32][HP]
    float* const simb = W2s + 32 * HP;                   // [AC][LBP]: every atom's own column block
    float* const zp = simb + AC * LBP;                   // [AC][HP]: pre, later d pre
    float* const dzb = zp + AC * HP;                     // [AC][HP]: z, then swish(pre), then d z
    float* const emol = dzb + AC * HP;                   // [MC][HP]: d e_g
    float* const vec = emol + MC * HP;                   // b1 | b2 | wh
    int* const info = (int*)(vec + 96);                  // [AC] off | len << 8
    int* const amol = info + AC;                         // [AC] molecule of the chunk
    int* const rpin = amol + AC;                         // [AC + 1] local edge offsets, by target
    int* const rpout = rpin + AC + 1;                    // [AC + 1] by source
    int* const ecin = rpout + AC + 1;                    // [EC] local source of every in-edge (-1: outside the chunk)
    int* const ecout = ecin + EC;                        // [EC]
    int* const mptr = ecout + EC;                        // [MC + 1] first atom of every molecule of the window
    int* const mein = mptr + MC + 1;                     // [MC + 1] first in-edge
    int* const meout = mein + MC + 1;                    // [MC + 1] first out-edge
    int* const ctl = meout + MC + 1;                     // [0] molecules of this chunk
    const int t = threadIdx.x, j = t & 31, slot = t >> 5;
    const int K = a.K, H = a.H, G = a.G;

    // ---- once: the weights.  Zero first (rows / lanes beyond H, G, K stay zero), then fill
    for (int i = t; i < KMAX * HP + 32 * HP; i += NT) lds[i] = 0.f;
    if (t < 96) vec[t] = 0.f;
    __syncthreads();
    for (int base = 0; base < H * K; base += NT * 8) {
        float tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = base + t + NT * u; tmp[u] = a.w1[i < H * K ? i : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + t + NT * u;
            if (i < H * K) { const int r = i / K, c = i - r * K; W1t[c * HP + r] = tmp[u]; }
        }
    }
    for (int i = t; i < G * H; i += NT) { const int r = i / H, c = i - r * H; W2s[r * HP + c] = a.w2[i]; }
    if (t < H && a.b1) vec[t] = a.b1[t];
    if (t < G && a.b2) vec[32 + t] = a.b2[t];
    if (t < G) vec[64 + t] = a.wh[t];
    const float bh = a.bh ? a.bh[0] : 0.f;
    const bool drop = a.drop_p > 0.f;
    const uint64_t seed = drop ? (uint64_t)a.rng[0] : 0, offset = drop ? (uint64_t)a.rng[1] : 0;
    const float invB = 1.f / (float)a.n_loss;

    // this workgroup's molecules: an equal share of a contiguous run
    const int64_t per = (a.n_mols + gridDim.x - 1) / gridDim.x;
    int64_t m_next = per * blockIdx.x;
    const int64_t m_hi = m_next + per < a.n_mols ? m_next + per : a.n_mols;

    // gradient accumulators that live across the chunks
    float accW1[CG];                                     // dW1[j][CG * slot + k]
#pragma unroll
    for (int k = 0; k < CG; ++k) accW1[k] = 0.f;
    float accW2[32];                                     // dW2[i = j][k]   (this slot's molecules)
#pragma unroll
    for (int k = 0; k < 32; ++k) accW2[k] = 0.f;
    float acc_b1 = 0.f, acc_b2 = 0.f, acc_wh = 0.f, acc_bh = 0.f, acc_loss = 0.f;

    __syncthreads();
    while (m_next < m_hi) {
        const int64_t m0 = m_next;
        // ---- the window: first atom and first edges of the next MC + 1 molecules (two dependent loads for all of them)
        if (t <= MC) {
            const int64_t m = m0 + t < a.n_mols ? m0 + t : a.n_mols;
            const int at = a.mol_ptr[m];
            mptr[t] = at;
            mein[t] = a.rin[at];
            meout[t] = a.rout[at];
        }
        __syncthreads();
        if (t == 0) {
            int nm = 0;
            const int lim = (int)(m_hi - m0 < MC ? m_hi - m0 : MC);
            while (nm < lim && mptr[nm + 1] - mptr[0] <= AC && mein[nm + 1] - mein[0] <= EC && meout[nm + 1] - meout[0] <= EC) ++nm;
            if (nm == 0) { nm = -1; atomicExch(a.status, 1); }          // one molecule beyond a chunk: skipped, reported
            ctl[0] = nm;
        }
        __syncthreads();
        int nm = ctl[0];
        if (nm < 0) {                                    // (block-uniform)
            if (t == 0) acc_loss = __builtin_nanf("");
            m_next = m0 + 1;
            __syncthreads();
            continue;
        }
        m_next = m0 + nm;
        const int a0 = mptr[0], A = mptr[nm] - a0;
        const int ein0 = mein[0], nin = mein[nm] - ein0, eout0 = meout[0], nout = meout[nm] - eout0;
        // ---- P1: per atom -- degree block, molecule, edge offsets
        if (t <= A) {
            const int n = a0 + (t < A ? t : A - 1);
            const int d = (int)a.deg[n];
            const int off = d >= 1 && d <= 4 ? (int)((a.blk_off >> (8 * (d - 1))) & 0xFF) : 0;
            const int len = d >= 1 && d <= 4 ? (int)((a.blk_len >> (8 * (d - 1))) & 0xFF) : 0;
            const int mol = a.atom_mol[n] - (int)m0;
            const int ri = a.rin[a0 + t] - ein0, ro = a.rout[a0 + t] - eout0;      // (t == A: the end of the last atom's edges)
            if (t < A) { info[t] = off | (len << 8); amol[t] = mol; }
            rpin[t] = ri; rpout[t] = ro;
        }
        __syncthreads();
        // ---- P2: the edges (local atom numbers) and every atom's own block of sim
        for (int i = t; i < nin; i += NT) { const int s = a.cin[ein0 + i] - a0; ecin[i] = (s >= 0 && s < A) ? s : -1; }
        for (int i = t; i < nout; i += NT) { const int s = a.cout[eout0 + i] - a0; ecout[i] = (s >= 0 && s < A) ? s : -1; }
        for (int base = 0; base < A * LB; base += NT * 13) {
            float tmp[13];
#pragma unroll
            for (int u = 0; u < 13; ++u) {
                const int i = base + t + NT * u, ic = i < A * LB ? i : A * LB - 1;
                const int at = ic / LB, c = ic - at * LB;
                const int off = info[at] & 0xFF, len = info[at] >> 8;
                tmp[u] = a.sim[(int64_t)(a0 + at) * a.ss + off + (c < len ? c : 0)];          // unconditional, masked below
            }
#pragma unroll
            for (int u = 0; u < 13; ++u) {
                const int i = base + t + NT * u;
                if (i < A * LB) {
                    const int at = i / LB, c = i - at * LB;
                    simb[at * LBP + c] = c < (info[at] >> 8) ? tmp[u] : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- P3a: z = W1[:, block] sim[block]
        for (int at = slot; at < A; at += 8) {
            const int off = info[at] & 0xFF, len = info[at] >> 8;
            float z = 0.f;
            for (int c = 0; c < len; ++c) z = fmaf(W1t[(off + c) * HP + j], simb[at * LBP + c], z);
            dzb[at * HP + j] = z;
        }
        __syncthreads();
        // ---- P3b: pre = b1 + sum over the in-edges; swish
        float sw[AC / 8];
#pragma unroll
        for (int q = 0; q < AC / 8; ++q) {
            const int at = slot + 8 * q;
            sw[q] = 0.f;
            if (at < A) {
                float p = vec[j];
                for (int e = rpin[at]; e < rpin[at + 1]; ++e) { const int s = ecin[e]; if (s >= 0) p += dzb[s * HP + j]; }
                zp[at * HP + j] = p;
                sw[q] = p * sigmoidf_(p);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < AC / 8; ++q) { const int at = slot + 8 * q; if (at < A) dzb[at * HP + j] = sw[q]; }
        __syncthreads();
        // ---- P4: per molecule -- pool, lin2, head, loss, and the way back to d e_g
        for (int g = slot; g < nm; g += 8) {
            const int b0 = mptr[g] - a0, b1_ = mptr[g + 1] - a0;
            float e = 0.f;
            for (int at = b0; at < b1_; ++at) e += dzb[at * HP + j];
            const float cnt = (float)(b1_ - b0);
            float emb = vec[32 + j] * cnt;
#pragma unroll
            for (int k = 0; k < 32; ++k) emb = fmaf(W2s[j * HP + k], lane_of(e, k), emb);
            const int64_t gi = m0 + g;
            if (a.emb && j < G) a.emb[gi * a.es + j] = emb;
            const bool counted = gi < a.n_loss;
            const float ks = (drop && j < G) ? keep_scale_of(seed, offset, (uint64_t)(counted ? gi : 0) * G + j, a.drop_p) : 1.f;
            const float wv = vec[64 + j];
            const float x = half_sum(j < G ? emb * ks * wv : 0.f) + bh;
            float d = 0.f;
            if (counted) {
                const float yv = a.y[gi];
                d = invB * (sigmoidf_(x) - yv);
                if (j == 0) {
                    a.pred[gi] = x;
                    acc_loss += invB * (fmaxf(x, 0.f) - x * yv + log1pf(expf(-fabsf(x))));      // torch's stable form
                    acc_bh += d;
                }
            }
            acc_wh = fmaf(d, emb * ks, acc_wh);
            const float demb = j < G ? d * wv * ks : 0.f;
            acc_b2 = fmaf(demb, cnt, acc_b2);
#pragma unroll
            for (int k = 0; k < 32; ++k) accW2[k] = fmaf(demb, lane_of(e, k), accW2[k]);
            float de = 0.f;                              // d e_g[j] = sum_i W2[i][j] d emb_i
#pragma unroll
            for (int i = 0; i < 32; ++i) de = fmaf(W2s[i * HP + j], lane_of(demb, i), de);
            emol[g * HP + j] = de;
        }
        __syncthreads();
        // ---- P5: d pre = d e_g * swish'(pre)   (in place of pre)
        for (int at = slot; at < A; at += 8) {
            const float p = zp[at * HP + j], sg = sigmoidf_(p);
            const float dp = emol[amol[at] * HP + j] * (sg * (1.f + p * (1.f - sg)));
            zp[at * HP + j] = dp;
            acc_b1 += dp;
        }
        __syncthreads();
        // ---- P6: d z = sum of d pre over the out-edges
        for (int at = slot; at < A; at += 8) {
            float v = 0.f;
            for (int e = rpout[at]; e < rpout[at + 1]; ++e) { const int s = ecout[e]; if (s >= 0) v += zp[s * HP + j]; }
            dzb[at * HP + j] = v;
        }
        __syncthreads();
        // ---- P7a: d sim[block] = W1[:, block]^T d z
        for (int i = t; i < A * LB; i += NT) {
            const int at = i / LB, c = i - at * LB;
            const int off = info[at] & 0xFF, len = info[at] >> 8;
            if (c < len) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) v = fmaf(W1t[(off + c) * HP + k], dzb[at * HP + k], v);
                a.gsim[(int64_t)(a0 + at) * a.gs + off + c] = v;
            }
        }
        // ---- P7b: dW1[j][col] += d z[j] sim[col]   (a thread's CG columns; an atom's block meets them or not)
        {
            const int c0 = CG * slot;
            for (int at = 0; at < A; ++at) {
                const int off = info[at] & 0xFF, len = info[at] >> 8;
                if (off < c0 + CG && off + len > c0) {
                    const float dzv = dzb[at * HP + j];
#pragma unroll
                    for (int k = 0; k < CG; ++k) {
                        const int c = c0 + k - off;
                        if (c >= 0 && c < len) accW1[k] = fmaf(dzv, simb[at * LBP + c], accW1[k]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- the workgroup's slab: dW1 as it is; the rest summed over the eight row slots in slot order
    float* const slab = a.slab + (size_t)blockIdx.x * a.slab_stride;
    if (j < H) {
#pragma unroll
        for (int k = 0; k < CG; ++k) { const int c = CG * slot + k; if (c < K) slab[j * K + c] = accW1[k]; }
    }
    float* const red = simb;                             // [8][32 * 36] >= what follows (AC * LBP floats = 6784 < 8 * 1152: use zp too)
    // (simb, zp and dzb are contiguous: 6784 + 2 * 4224 floats)
#pragma unroll
    for (int k = 0; k < 32; ++k) red[(slot * 32 + j) * 36 + k] = accW2[k];
    red[(slot * 32 + j) * 36 + 32] = acc_b1;
    red[(slot * 32 + j) * 36 + 33] = acc_b2;
    red[(slot * 32 + j) * 36 + 34] = acc_wh;
    red[(slot * 32 + j) * 36 + 35] = (j == 0) ? acc_bh : 0.f;
    if (j == 0) vec[slot] = acc_loss;                    // (b1's copy is no longer needed)
    __syncthreads();
    float* const so = slab + H * K;
    for (int i = t; i < 32 * 36; i += NT) {
        const int r = i / 36, c = i - r * 36;
        float v = 0.f;
        for (int s8 = 0; s8 < 8; ++s8) v += red[(s8 * 32 + r) * 36 + c];
        if (c < 32) so[32 + r * 32 + c] = v;             // W2 [32][32]
        else if (c == 32) so[r] = v;                     // b1 [32]
        else if (c == 33) so[32 + 1024 + r] = v;         // b2 [32]
        else if (c == 34) so[32 + 1024 + 32 + r] = v;    // wh [32]
        else if (r == 0) so[32 + 1024 + 64] = v;         // bh
    }
    if (t == 0) {
        float v = 0.f;
        for (int s8 = 0; s8 < 8; ++s8) v += vec[s8];
        so[32 + 1024 + 65] = v;                          // loss (already divided by B)
    }
}

constexpr int SLAB_TAIL = 32 + 1024 + 32 + 32 + 2;      // floats behind dW1 in a slab

struct RedArgs {
    const float* slab; int stride, count;
    int K, H, G;
    float *gw1, *gb1, *gw2, *gb2, *gwh, *gbh, *loss;
    float drop_p; int64_t* rng; int64_t* rng_used;
    int* status;
};

// element e of the reduced slab -> its destination; 32 elements per block, eight slab parts per element, fixed order
__global__ void __launch_bounds__(256) tail_reduce_kernel(RedArgs a) {
    __shared__ float part[8][32];
    const int total = a.H * a.K + SLAB_TAIL;
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), p = threadIdx.x >> 5;
    const int ec = e < total ? e : total - 1;
    const int per = (a.count + 7) / 8;
    const int b0 = p * per, b1 = (b0 + per < a.count) ? b0 + per : a.count;
    float s = 0.f;
    for (int b = b0; b < b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = a.slab[(int64_t)(b + u < b1 ? b + u : b1 - 1) * a.stride + ec];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (b + u < b1) s += v[u];
    }
    part[p][threadIdx.x & 31] = s;
    __syncthreads();
    if (p == 0 && e < total) {
        float v = part[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += part[k][threadIdx.x];
        const int hk = a.H * a.K;
        if (e < hk) { if (a.gw1) a.gw1[e] = v; }
        else {
            const int r = e - hk;
            if (r < 32) { if (a.gb1 && r < a.H) a.gb1[r] = v; }
            else if (r < 32 + 1024) { const int i = (r - 32) >> 5, k = (r - 32) & 31; if (a.gw2 && i < a.G && k < a.H) a.gw2[i * a.H + k] = v; }
            else if (r < 32 + 1024 + 32) { const int i = r - 32 - 1024; if (a.gb2 && i < a.G) a.gb2[i] = v; }
            else if (r < 32 + 1024 + 64) { const int i = r - 32 - 1024 - 32; if (a.gwh && i < a.G) a.gwh[i] = v; }
            else if (r == 32 + 1024 + 64) { if (a.gbh) a.gbh[0] = v; }
            else {
                const int bad = *a.status;                // a molecule beyond a chunk: NaN, and the word is left zero for the next call
                a.loss[0] = bad ? __builtin_nanf("") : v;
                if (bad) *a.status = 0;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.drop_p > 0.f) {      // the generator advances once per step, as after mkgnn_bce_head_fused
        const int64_t seed = a.rng[0], offset = a.rng[1];
        a.rng_used[0] = seed; a.rng_used[1] = offset;
        a.rng[1] = offset + 1;
    }
}

static int grid_for(int64_t n_mols) {
    int64_t nb = (n_mols + 7) / 8;                       // ~8 molecules (two chunks) per workgroup ...
    if (nb > 512) nb = 512;                              // ... and no more slabs than the reduction reads in ~5 us
    return (int)(nb < 1 ? 1 : nb);
}

}  // namespace tail
}  // namespace mkgnn

using namespace mkgnn;

extern "C" {

int mkgnn_tail_supported(int32_t K, int32_t H, int32_t G, const int32_t num_kernels[MKGNN_MAX_DEGREE]) {
    if (!num_kernels || K < 1 || K > tail::KMAX || H < 1 || H > 32 || G < 1 || G > 32) return 0;
    int sum = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        if (num_kernels[i] < 0 || num_kernels[i] > tail::LB) return 0;
        sum += num_kernels[i];
    }
    return sum == K ? 1 : 0;
}

size_t mkgnn_tail_workspace_bytes(int32_t K, int32_t H, int64_t n_mols) {
    if (K < 1 || H < 1 || n_mols < 1) return 0;
    return 16 + (size_t)tail::grid_for(n_mols) * (size_t)((H * K + tail::SLAB_TAIL + 3) / 4 * 4) * 4;
}

int mkgnn_tail_fused(const mkgnn_tail_args* p, void* workspace, size_t workspace_bytes, void* stream) {
    const char* who = "mkgnn_tail_fused";
    if (!p) return api_fail("%s: null argument", who);
    const mkgnn_readout_params& ro = p->readout;
    if (!mkgnn_tail_supported(ro.F, ro.H, ro.G, p->num_kernels))
        return api_fail("%s: K=%d H=%d G=%d outside the fused tail (K <= %d, blocks <= %d, H, G <= 32)", who, ro.F, ro.H, ro.G, tail::KMAX, tail::LB);
    if (p->n_atoms < 1 || p->n_mols < 1 || p->n_loss_mols < 1 || p->n_loss_mols > p->n_mols) return api_fail("%s: bad sizes", who);
    if (!p->sim || !p->degree || !p->in_rowptr || !p->in_col || !p->out_rowptr || !p->out_col || !p->mol_ptr || !p->atom_mol ||
        !ro.lin1_weight || !ro.lin2_weight || !p->head_weight || !p->target || !p->pred || !p->loss || !p->grad_sim)
        return api_fail("%s: null pointer", who);
    if (p->sim_stride < ro.F || p->grad_sim_stride < ro.F) return api_fail("%s: bad strides", who);
    if (p->dropout_p < 0.f || p->dropout_p >= 1.f) return api_fail("%s: dropout probability %g outside [0, 1)", who, (double)p->dropout_p);
    if (p->dropout_p > 0.f && (!p->rng_state || !p->rng_used)) return api_fail("%s: dropout needs rng_state and rng_used", who);
    if (p->emb && p->emb_stride < ro.G) return api_fail("%s: bad emb stride", who);
    const size_t need = mkgnn_tail_workspace_bytes(ro.F, ro.H, p->n_mols);
    if (!workspace || workspace_bytes < need) return api_fail("%s: workspace of %zu bytes, need %zu", who, workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    tail::Args a{};
    a.sim = p->sim; a.ss = p->sim_stride; a.deg = p->degree;
    a.rin = p->in_rowptr; a.cin = p->in_col; a.rout = p->out_rowptr; a.cout = p->out_col;
    a.mol_ptr = p->mol_ptr; a.atom_mol = p->atom_mol;
    a.n_atoms = p->n_atoms; a.n_mols = p->n_mols; a.n_loss = p->n_loss_mols;
    a.w1 = ro.lin1_weight; a.b1 = ro.lin1_bias; a.w2 = ro.lin2_weight; a.b2 = ro.lin2_bias;
    a.wh = p->head_weight; a.bh = p->head_bias; a.y = p->target;
    a.K = ro.F; a.H = ro.H; a.G = ro.G;
    int off = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        a.blk_off |= (uint32_t)off << (8 * i);
        a.blk_len |= (uint32_t)p->num_kernels[i] << (8 * i);
        off += p->num_kernels[i];
    }
    a.drop_p = p->dropout_p; a.rng = p->rng_state;
    a.emb = p->emb; a.es = p->emb_stride;
    a.pred = p->pred; a.gsim = p->grad_sim; a.gs = p->grad_sim_stride;
    a.status = (int*)workspace;
    a.slab = (float*)((char*)workspace + 16);
    a.slab_stride = (ro.H * ro.F + tail::SLAB_TAIL + 3) / 4 * 4;
    const int nb = tail::grid_for(p->n_mols);
    hipError_t e = hipSuccess;
    const size_t lds_bytes = (size_t)tail::lds_floats() * 4 + (size_t)tail::lds_ints() * 4;
    static PerDeviceOnce attr_set;
    if (const int slot = attr_set.pending(); slot >= 0) {
        e = hipFuncSetAttribute((const void*)tail::tail_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return api_hip_fail(who, e);
        attr_set.set(slot);
    }
    tail::tail_fused_kernel<<<nb, tail::NT, lds_bytes, st>>>(a);
    e = hipGetLastError();
    if (e != hipSuccess) return api_hip_fail(who, e);
    tail::RedArgs r{};
    r.slab = a.slab; r.stride = a.slab_stride; r.count = nb; r.K = ro.F; r.H = ro.H; r.G = ro.G;
    r.gw1 = p->grad_lin1_weight; r.gb1 = p->grad_lin1_bias; r.gw2 = p->grad_lin2_weight; r.gb2 = p->grad_lin2_bias;
    r.gwh = p->grad_head_weight; r.gbh = p->grad_head_bias; r.loss = p->loss;
    r.drop_p = p->dropout_p; r.rng = p->rng_state; r.rng_used = p->rng_used; r.status = a.status;
    const int total = ro.H * ro.F + tail::SLAB_TAIL;
    tail::tail_reduce_kernel<<<(total + 31) / 32, 256, 0, st>>>(r);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : api_hip_fail(who, e);
}

}  // extern "C"
