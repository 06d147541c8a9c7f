// The tail of a training step (round 6): everything behind the last kernel convolution --
//
//   h = propagate(sim_sc)                                   reference KernelLayer.py:119-123
//   emb_g = pool_g( lin2( swish( lin1(h) ) ) )              reference MolKGNNNet.py:144-146
//   loss = BCEWithLogits( ffn( dropout(emb) ), y )          reference model.py:147-150, 169, 190-198; data.py:37
//
// -- forward AND backward in FOUR launches instead of nine (mkgnn_tail_fused, kgnn_readout.hip).  The loss is a mean over
// molecules, so d loss / d logit_g = (sigmoid(logit_g) - y_g) / B needs nothing but the molecule's own logit, and the chain back
// to d loss / d z can be taken while the molecule is still in LDS.  Rounds 3-5 ran: project, propagate, pool, head, head reduce |
// molecule gradients, propagate^T, project^T + dW1, slab reduce -- 102 us of a 725 us step at batch 4096, about half of it launch
// latency (a dependent kernel costs ~5 us in a captured graph on this chip whatever it does).  Now:
//     z = W1[:, block] sim[block]                                        block_project_mfma_kernel        (matrix cores, as before)
//     THIS FILE: per chunk of whole molecules, on the H-wide rows --     tail_middle_kernel
//         pre = b1 + sum of z over the in-edges;  e_g = sum_n swish(pre_n);  emb_g = W2 e_g + |g| b2;  logit, loss, d logit;
//         d emb, d e_g;  d pre = d e_g * swish'(pre);  d z = sum of d pre over the out-edges;  partial dW2, db1, db2, d head
//     d sim[block] = W1[:, block]^T d z,  dW1 += d z (x) sim              block_project_bwd_mfma_kernel    (matrix cores, as before)
//     all partial slabs -> the gradients, the loss                       slab_reduce_kernel
// (A first version did the two projections here too, on the vector pipe out of LDS: 530 us -- three 98 MFLOP products with an
// LDS read per multiply-add at one wave per SIMD.  They stay on the matrix cores.)
//
// A workgroup takes a contiguous run of whole molecules and walks it in CHUNKS that fit LDS (<= AC atoms, <= EC edges each way,
// <= MC molecules): bulk loads of the chunk's indices and z rows, block-wide barriers between the phases.  Molecules never
// share atoms or edges, so a chunk needs nothing from outside.  No float atomics: every sum runs in a fixed order, the
// per-workgroup partial gradients go to slabs that the reduction adds up in block order.  The dropout mask of the head comes
// from the same counter-based generator, element for element, as mkgnn_bce_head_fused (kgnn_philox.h).
//
// Limits (mkgnn_tail_supported; the host falls back to the nine launches otherwise): H <= 32, G <= 32, no dropout inside the
// readout, and -- checked per batch by the host, which knows the molecule sizes -- no single molecule beyond a chunk (128
// atoms, 512 edges).  A molecule that breaks the last promise is skipped and the loss comes back NaN: loud, not wrong.
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
constexpr int AC = MKGNN_TAIL_MAX_ATOMS, EC = MKGNN_TAIL_MAX_EDGES, MC = 12;      // chunk capacity: atoms, edges (each way), molecules
constexpr int MG_MIN = 1, MG_MAX = 8;   // molecules per group (the unit of work distribution): tail_group_size below
constexpr int HP = 36;                  // LDS pitch of the 32-wide rows: 16-byte aligned rows, 4 banks apart
constexpr int WP = 33;                  // ... of W2's rows (read one float per lane: odd, conflict-free)
typedef mkgnn_f32x4 f32x4;

// sigmoid on the transcendental unit: v_exp_f32 and v_rcp_f32 (1 ulp each) instead of the library's expf and an IEEE division --
// ~6 instructions for ~60.  This kernel evaluates two of them per (atom, hidden unit) and, measured by compiling the phases out,
// was BOUND by them (31 of its 64 us).  Relative error <= 2^-22 + |v| 2^-23: 2e-6 at |v| = 16, against the 1e-5 the readout is
// held to (tests/test_tail.py: against float64 autograd of the reference's formula).
__device__ __forceinline__ float sigmoidf_(float v) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * v);      // e^-v  (inf for v << 0: the reciprocal is then 0)
    return __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ float half_sum(float v) {       // xor tree over the 32 lanes of a row slot
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float lane_of(float v, int k) { return __shfl(v, (int)(threadIdx.x & 32) | k, 64); }    // lane k of my row slot

// out[u] = sum over the edges of atom at0 + 32 u of rows[col][4 l .. 4 l + 3]  (rp: local edge offsets, ec: local atom of every
// edge, -1 = none; a thread is (atom slot, lane l of 8): 16 bytes of a 32-wide row).  The first four edges of all NA atoms in
// one batch of reads -- offsets, then columns, then rows -- and a serial tail for the rare atom with more (atoms outside the
// degree buckets); an atom beyond A contributes nothing.  Fixed order: edge by edge.
template <int NA>
__device__ __forceinline__ void gather_rows(const int* rp, const int* ec, const float* rows, int at0, int A, int l, f32x4 (&out)[NA]) {
    int lo[NA], hi[NA], col[NA][4];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int at = at0 + 32 * u, ac = at < A ? at : A - 1;
        lo[u] = rp[ac]; hi[u] = at < A ? rp[ac + 1] : lo[u];
    }
#pragma unroll
    for (int u = 0; u < NA; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) col[u][k] = ec[lo[u] + k < hi[u] ? lo[u] + k : (hi[u] > lo[u] ? lo[u] : 0)];
    f32x4 v[NA][4];
#pragma unroll
    for (int u = 0; u < NA; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = *(const f32x4*)(rows + ((unsigned)col[u][k] < (unsigned)A ? col[u][k] : 0) * HP + 4 * l);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (lo[u] + k < hi[u] && col[u][k] >= 0) s += v[u][k];
        for (int e = lo[u] + 4; e < hi[u]; ++e) { const int c = ec[e]; if (c >= 0) s += *(const f32x4*)(rows + c * HP + 4 * l); }
        out[u] = s;
    }
}

// Diagnostics (always compiled, off unless mkgnn_debug_set_tail_stamps gave a buffer): cycle totals of the middle kernel's phases,
// thread 0 of every workgroup -> [block][16] (tools/tail_stamps.py)
__device__ unsigned long long* g_tail_stamps = nullptr;
#define TAIL_PHASE(i) do { if (stamps) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - t_ph; t_ph = t_; } } while (0)

constexpr int LDS_FLOATS = 32 * WP + 2 * AC * HP + 3 * MC * HP + 96;
constexpr int LDS_INTS = AC + 2 * (AC + 1) + 2 * EC + 3 * (MC + 1) + 4;

__global__ void __launch_bounds__(NT, 3) tail_middle_kernel(TailMidArgs a) {
    __shared__ __align__(16) float lds[LDS_FLOATS + LDS_INTS];
    float* const W2s = lds;                              // [32][HP]
    float* const zp = W2s + 32 * WP;                     // [AC][HP]: pre, later d pre   (32 * 33 floats = 4 224 bytes: 16-byte aligned)
    float* const dzb = zp + AC * HP;                     // [AC][HP]: z, then swish(pre), then d z
    float* const emol = dzb + AC * HP;                   // [MC][HP]: d e_g
    float* const emol_e = emol + MC * HP;                // [MC][HP]: e_g
    float* const emol_d = emol_e + MC * HP;              // [MC][HP]: d emb_g
    float* const vec = emol_d + MC * HP;                 // b1 | b2 | wh
    int* const amol = (int*)(vec + 96);                  // [AC] molecule of the chunk
    int* const rpin = amol + AC;                         // [AC + 1] local edge offsets, by target
    int* const rpout = rpin + AC + 1;                    // [AC + 1] by source
    int* const ecin = rpout + AC + 1;                    // [EC] local source of every in-edge (-1: outside the chunk)
    int* const ecout = ecin + EC;                        // [EC]
    int* const mptr = ecout + EC;                        // [MC + 1] first atom of every molecule of the window
    int* const mein = mptr + MC + 1;                     // [MC + 1] first in-edge
    int* const meout = mein + MC + 1;                    // [MC + 1] first out-edge
    int* const ctl = meout + MC + 1;                     // [0] molecules of this chunk
    const int t = threadIdx.x, j = t & 31, slot = t >> 5;      // the per-molecule phase: (row slot of 8, hidden unit)
    const int l = t & 7, as = t >> 3;                          // the per-atom phases: (atom slot of 32, 16-byte lane of 8)
    const int H = a.H, G = a.G;
    unsigned long long* const stamps = g_tail_stamps;
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_ph = stamps ? __builtin_readcyclecounter() : 0ull;

    // ---- once: the weights.  Zero first (rows / lanes beyond H, G stay zero), then fill
    // (every load first -- unconditional, clamped -- then the stores: written as "lds[i] = ok ? w[i] : 0" in a loop the compiler keeps
    // one load in flight at a time, and this kernel is nothing but latency)
    float w2r[4], vr = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = t + NT * u; w2r[u] = a.w2[i < G * H ? i : 0]; }
    {
        const float* vp = t < 32 ? a.b1 : (t < 64 ? a.b2 : a.wh);
        const int lim = t < 32 ? H : G, k = t & 31;
        if (t < 96 && vp && k < lim) vr = vp[k];         // (three short segments: one load each, no chain)
    }
    const float bh = a.bh ? a.bh[0] : 0.f;
    const bool drop = a.drop_p > 0.f;
    const uint64_t seed = drop ? (uint64_t)a.rng[0] : 0, offset = drop ? (uint64_t)a.rng[1] : 0;
    const float invB = 1.f / (float)a.n_loss;
    for (int i = t; i < 32 * WP; i += NT) W2s[i] = 0.f;
    if (t < 96) vec[t] = vr;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = t + NT * u; if (i < G * H) { const int r = i / H, c = i - r * H; W2s[r * WP + c] = w2r[u]; } }

    // this workgroup's molecules: GROUPS of MG consecutive molecules, group g to workgroup g mod gridDim.x -- and gridDim.x a
    // function of n_loss alone (tail_middle_blocks): which molecules share a workgroup, a chunk and a slab does not depend on
    // how many padding molecules follow the real ones, so a batch padded to a fixed shape (molkgnn_amd.padding) gives bit for
    // bit the loss and the gradients of the unpadded batch (the padding molecules add exact zeros)
    const int MG = a.mg;
    const int64_t n_groups = (a.n_mols + MG - 1) / MG;
    int64_t grp = blockIdx.x;
    int64_t m_next = grp * MG;
    int64_t m_hi = m_next + MG < a.n_mols ? m_next + MG : a.n_mols;
    if (grp >= n_groups) m_next = m_hi = 0;

    float accW2[4] = {0.f, 0.f, 0.f, 0.f};               // dW2[i][k], element t + 256 u of the [32][32] image: summed per chunk from the
                                                         // molecules' (d emb, e) pairs kept in LDS (32 accumulators per thread cost 60 VGPRs)
    f32x4 acc_b1 = {0.f, 0.f, 0.f, 0.f};                  // d b1[4 l .. 4 l + 3], this atom slot's atoms
    float acc_b2 = 0.f, acc_wh = 0.f, acc_bh = 0.f, acc_loss = 0.f;

    __syncthreads();
    TAIL_PHASE(0);                                       // prologue
    while (true) {
        if (m_next >= m_hi) {                            // (block-uniform) this group is done: the next one of this workgroup
            grp += gridDim.x;
            if (grp >= n_groups) break;
            m_next = grp * MG;
            m_hi = m_next + MG < a.n_mols ? m_next + MG : a.n_mols;
        }
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
            ctl[0] = nm;
        }
        __syncthreads();
        TAIL_PHASE(1);                                   // window
        const int nm = ctl[0];
        if (nm == 0) {                                   // (block-uniform) one molecule beyond a chunk: skipped, reported as NaN.
            // Its atoms still get a d z row (zero): the projection kernel behind this one reads every row.
            const int a0 = mptr[0], A = mptr[1] - a0;
            for (int i = t; i < A * 32; i += NT) a.dz[(int64_t)a0 * 32 + i] = 0.f;
            if (t == 0) acc_loss = __builtin_nanf("");
            m_next = m0 + 1;
            __syncthreads();
            continue;
        }
        m_next = m0 + nm;
        const int a0 = mptr[0], A = mptr[nm] - a0;
        const int ein0 = mein[0], nin = mein[nm] - ein0, eout0 = meout[0], nout = meout[nm] - eout0;
        // ---- P1: per atom -- molecule, edge offsets; the z rows (one coalesced 128-byte row per slot and pass)
        if (t <= A) {
            const int n = a0 + (t < A ? t : A - 1);
            const int mol = a.atom_mol[n] - (int)m0;
            const int ri = a.rin[a0 + t] - ein0, ro = a.rout[a0 + t] - eout0;      // (t == A: the end of the last atom's edges)
            if (t < A) amol[t] = mol;
            rpin[t] = ri; rpout[t] = ro;
        }
        {
            f32x4 zr[AC / 32];
#pragma unroll
            for (int q = 0; q < AC / 32; ++q) { const int at = as + 32 * q; zr[q] = *(const f32x4*)(a.z + (int64_t)(a0 + (at < A ? at : A - 1)) * 32 + 4 * l); }
#pragma unroll
            for (int q = 0; q < AC / 32; ++q) { const int at = as + 32 * q; if (at < A) *(f32x4*)(dzb + at * HP + 4 * l) = zr[q]; }
        }
        {
            int ci[EC / NT], co[EC / NT];                // (all four loads in flight: EC / NT = 2 edges each way per thread)
#pragma unroll
            for (int u = 0; u < EC / NT; ++u) {
                const int i = t + NT * u;
                ci[u] = a.cin[ein0 + (i < nin ? i : 0)];
                co[u] = a.cout[eout0 + (i < nout ? i : 0)];
            }
#pragma unroll
            for (int u = 0; u < EC / NT; ++u) {
                const int i = t + NT * u;
                if (i < nin) { const int s = ci[u] - a0; ecin[i] = (s >= 0 && s < A) ? s : -1; }
                if (i < nout) { const int s = co[u] - a0; ecout[i] = (s >= 0 && s < A) ? s : -1; }
            }
        }
        __syncthreads();
        TAIL_PHASE(2);                                   // chunk loads
#if defined(MKGNN_TAIL_ABLATE) && MKGNN_TAIL_ABLATE == 1      // (timing experiment: the loads and the stores only)
        for (int at = slot; at < A; at += 8) a.dz[(int64_t)(a0 + at) * 32 + j] = dzb[at * HP + j];
        __syncthreads();
        continue;
#endif
        // ---- P3: pre = b1 + sum of z over the in-edges; swish
        f32x4 sw[AC / 32];
        {
            // (all of a pass's index reads, then all of its row reads, then the sums: written atom by atom the two dependent LDS
            // round trips of every edge were paid one after the other; sixteen bytes per lane: a quarter of the LDS instructions)
            const f32x4 b1v = *(const f32x4*)(vec + 4 * l);
            f32x4 p[AC / 32];
            gather_rows<AC / 32>(rpin, ecin, dzb, as, A, l, p);
#pragma unroll
            for (int u = 0; u < AC / 32; ++u) {
                const int at = as + 32 * u;
                const f32x4 pv = b1v + p[u];
                sw[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (at < A) {
                    *(f32x4*)(zp + at * HP + 4 * l) = pv;
#pragma unroll
                    for (int c = 0; c < 4; ++c) sw[u][c] = pv[c] * sigmoidf_(pv[c]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < AC / 32; ++q) { const int at = as + 32 * q; if (at < A) *(f32x4*)(dzb + at * HP + 4 * l) = sw[q]; }
        __syncthreads();
        TAIL_PHASE(3);                                   // P3
        // ---- P4: per molecule -- pool, lin2, head, loss, and the way back to d e_g
#if defined(MKGNN_TAIL_ABLATE) && MKGNN_TAIL_ABLATE == 2      // (timing experiment: no per-molecule phase)
        for (int g = slot; g < nm; g += 8) { emol[g * HP + j] = 1.f; emol_e[g * HP + j] = 1.f; emol_d[g * HP + j] = 1.f; }
        for (int g = slot; false && g < nm; g += 8) {
#else
        for (int g = slot; g < nm; g += 8) {
#endif
            const int b0 = mptr[g] - a0, b1_ = mptr[g + 1] - a0;
            float e = 0.f;
            for (int at = b0; at < b1_; at += 8) {       // (eight reads in flight; the order of the sum stays atom by atom)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = dzb[(at + u < b1_ ? at + u : b1_ - 1) * HP + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (at + u < b1_) e += v[u];
            }
            const float cnt = (float)(b1_ - b0);
            float emb = vec[32 + j] * cnt;
#pragma unroll
            for (int k = 0; k < 32; ++k) emb = fmaf(W2s[j * WP + k], lane_of(e, k), emb);
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
            float de = 0.f;                              // d e_g[j] = sum_i W2[i][j] d emb_i
#pragma unroll
            for (int i = 0; i < 32; ++i) de = fmaf(W2s[i * WP + j], lane_of(demb, i), de);
            emol[g * HP + j] = de;
            emol_e[g * HP + j] = e;
            emol_d[g * HP + j] = demb;
        }
        __syncthreads();
        TAIL_PHASE(4);                                   // P4
#pragma unroll
        for (int u = 0; u < 4; ++u) {                    // dW2[i][k] += sum over the chunk's molecules of d emb_g[i] e_g[k]
            const int i = (t + NT * u) >> 5, k = (t + NT * u) & 31;
            for (int g = 0; g < nm; ++g) accW2[u] = fmaf(emol_d[g * HP + i], emol_e[g * HP + k], accW2[u]);
        }
        // ---- P5: d pre = d e_g * swish'(pre)   (in place of pre)
        {
            f32x4 p[AC / 32], de[AC / 32];
#pragma unroll
            for (int u = 0; u < AC / 32; ++u) {
                const int at = as + 32 * u, ac = at < A ? at : A - 1;
                p[u] = *(const f32x4*)(zp + ac * HP + 4 * l);
                de[u] = *(const f32x4*)(emol + amol[ac] * HP + 4 * l);
            }
#pragma unroll
            for (int u = 0; u < AC / 32; ++u) {
                const int at = as + 32 * u;
                if (at < A) {
                    f32x4 dp;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float sg = sigmoidf_(p[u][c]); dp[c] = de[u][c] * (sg * (1.f + p[u][c] * (1.f - sg))); }
                    *(f32x4*)(zp + at * HP + 4 * l) = dp;
                    acc_b1 += dp;
                }
            }
        }
        __syncthreads();
        TAIL_PHASE(5);                                   // dW2 + P5
        // ---- P6: d z = sum of d pre over the out-edges, straight to memory (a coalesced row per slot and pass)
        {
            f32x4 v[AC / 32];
            gather_rows<AC / 32>(rpout, ecout, zp, as, A, l, v);
#pragma unroll
            for (int u = 0; u < AC / 32; ++u) {
                const int at = as + 32 * u;
                if (at < A) *(f32x4*)(a.dz + (int64_t)(a0 + at) * 32 + 4 * l) = v[u];
            }
        }
        // (the LDS images are free for the next chunk once everybody has READ them: a barrier that does not wait for the d z
        // stores -- __syncthreads() would, a full memory round trip per chunk)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        TAIL_PHASE(6);                                   // P6
    }
    __syncthreads();
    TAIL_PHASE(7);

    // ---- the workgroup's slab: everything summed over the eight row slots in slot order
    float* const so = a.slab + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int u = 0; u < 4; ++u) so[TAIL_W2 + t + NT * u] = accW2[u];
    float* const red = zp;                               // [8 slots][32][4]: b2, wh, bh of the per-molecule phase
    float* const redb = dzb;                             // [32 atom slots][32]: b1 of the per-atom phases
    red[(slot * 32 + j) * 4 + 1] = acc_b2;
    red[(slot * 32 + j) * 4 + 2] = acc_wh;
    red[(slot * 32 + j) * 4 + 3] = (j == 0) ? acc_bh : 0.f;
    *(f32x4*)(redb + as * 32 + 4 * l) = acc_b1;
    if (j == 0) vec[slot] = acc_loss;                    // (b1's copy is no longer needed)
    __syncthreads();
    if (t < 128) {
        const int r = t >> 2, c = t & 3;
        float v = 0.f;
        if (c == 0) {
#pragma unroll
            for (int s32 = 0; s32 < 32; ++s32) v += redb[s32 * 32 + r];
            so[TAIL_B1 + r] = v;
        } else {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) v += red[(s8 * 32 + r) * 4 + c];
            if (c == 1) so[TAIL_B2 + r] = v;
            else if (c == 2) so[TAIL_WH + r] = v;
            else if (r == 0) so[TAIL_BH] = v;
        }
    }
    if (t == 0) {
        float v = 0.f;
        for (int s8 = 0; s8 < 8; ++s8) v += vec[s8];
        so[TAIL_LOSS] = v;                               // (already divided by B)
    }
    TAIL_PHASE(8);                                       // epilogue
    if (stamps && t == 0)
        for (int i = 0; i < 12; ++i) stamps[(size_t)blockIdx.x * 16 + i] = ph[i];
}

}  // namespace tail

// Molecules per group and workgroups, both functions of the number of REAL molecules alone (padding invariance: see the kernel).
// Three workgroups fit a CU (52 KB of LDS each), and a workgroup's time is its group's: as many groups as there are slots -- up
// to TAIL_MAX_BLOCKS = 3 x 256 -- of as few molecules as that allows.  Round 6 measured at 4 096 molecules: 8 per group (512
// workgroups, two per CU) 42 us; 7 (586) and 6 (683) 36-37; 5 (820: a second round of workgroups) 43.
int tail_group_size(int64_t n_loss_mols) {
    const int64_t g = (n_loss_mols + TAIL_MAX_BLOCKS - 1) / TAIL_MAX_BLOCKS;
    return (int)(g < tail::MG_MIN ? tail::MG_MIN : (g > tail::MG_MAX ? tail::MG_MAX : g));
}
int tail_middle_blocks(int64_t n_loss_mols) {
    const int mg = tail_group_size(n_loss_mols);
    int64_t nb = (n_loss_mols + mg - 1) / mg;                  // one group of real molecules per workgroup (padding molecules' groups wrap around),
    if (nb > TAIL_MAX_BLOCKS) nb = TAIL_MAX_BLOCKS;            // and no more slabs than the reduction reads in a few batches
    return (int)(nb < 1 ? 1 : nb);
}

extern "C" int mkgnn_debug_set_tail_stamps(void* device_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(tail::g_tail_stamps), &device_ptr, sizeof(void*));
}

hipError_t launch_tail_middle(const TailMidArgs& a, int nb, hipStream_t st) {
    if (a.mg < tail::MG_MIN || a.mg > tail::MG_MAX || a.mg > tail::MC) return hipErrorInvalidValue;
    tail::tail_middle_kernel<<<nb, tail::NT, 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace mkgnn
