// MFMA backward kernel of the kernel convolution (gfx950, exact-fp32 v_mfma_f32_16x16x4_f32):
// the gradient with respect to the unit feature rows.
//
// It is a contraction with a 0/1-masked coefficient matrix
//     P_sb[n, l] = coef[n, l]  if the chosen permutation of (kernel l, atom n) matched neighbour
//                              slot s to support b, else 0          (coef = dL/dsc * w_s / (W d))
//   rows:  g_xhat[n, slot s, :] = sum_b sum_l P_sb[n, l] * unit_support[l, b, :]     (K = kernels)
// (plus the centre rows with coef * w_c d / w_s).  Written densely it costs d times the sparse FMA
// count, but it runs on the matrix pipe with the masked operand built in registers (one v_cndmask
// per MFMA) instead of one LDS read per two FMAs: measured 130 us against 185 us per N-hop layer
// (batch 4096) for the LDS/VALU kernel of kgnn_bwd.hip, which stays as the fallback for wider
// kernel banks.  The transposed product (bank gradients, K = atoms) was built the same way and
// measured SLOWER than the LDS/VALU bank kernel at these sizes (291 us vs 227 us: its per-tile
// operand gathers dominate), so the bank gradients stay on kc_backward_bank_lds.
//
// Like the forward, the kernel is wave-autonomous: no block barrier after the one-time bank copy;
// every coefficient comes straight from global memory in MFMA operand layout.  All loads are
// unconditional on clamped addresses and masked afterwards (a load under a lane-dependent branch
// ends its basic block with a full wait and serialises the round trips).
#include "kgnn_launch.h"

namespace mkgnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic cycle stamps (tools/bwd_stamp_probe.py), compiled in only with make STAMPS=1.
__device__ unsigned long long* g_rows_stamp_buffer = nullptr;
#ifdef MKGNN_BWD_STAMPS
#define ROWS_STAMP(slot)                                                                         \
    do {                                                                                         \
        if (stamps && lane == 0 && (slot) < 32) stamps[(slot)] = __builtin_readcyclecounter();   \
    } while (0)
#else
#define ROWS_STAMP(slot) do { (void)stamps; (void)(slot); } while (0)
#endif

__host__ __device__ constexpr int bwd_lq(int d) { return d == 1 ? 3 : (d == 2 ? 5 : (d == 3 ? 8 : 13)); }   // 4-kernel groups held per lane
// Feature slices: a BLOCK works on one slice of the feature tiles (the slices are independent: no reduction), its
// four waves on four different atom tiles.  Degree 4 has few atoms and 17 masked products per kernel group -- one
// unsliced wave per tile left three quarters of the SIMDs idle; and a block keeps only its slice of the bank in LDS
// (48 / 38 / 19 KB instead of 112 / 54 / 27 KB for degree 4 / 3 / 2), which shortens the copy and leaves room for
// the bank-gradient kernel that runs next to this one inside a captured graph.
__host__ __device__ constexpr int bwd_fsplit(int d, int kc) { return kc < 4 ? 1 : (d == 1 ? 1 : (d == 4 ? 4 : 2)); }
// LDS row stride of a slice of fpw floats: odd multiple of 16 floats, so that the four kernel rows a wave reads
// together (64 lanes x 4 bytes) spread over all 32 banks
__host__ __device__ constexpr int bwd_slice_stride(int fpw) { return (fpw % 32 == 16) ? fpw : fpw + 16; }

// ------------------------------------------------------------------ rows ---
// M = 16 atoms, N = 16 features (FT tiles), K = kernels.  A = masked coefficients (registers),
// B = unit kernel rows (LDS, b32 reads, conflict-free), D = contribution rows.
// (The three score-weight partials d sc / d theta_k are summed by the bank kernel, which visits every
// (atom, kernel) pair with one thread and has registers to spare; here they cost 3 LQ prefetch registers.)
template <int D, int KC, int NT>
__global__ void __launch_bounds__(NT, 2) kc_backward_rows_mfma(BwdArgs a) {
    constexpr int FP = 16 * KC;
    constexpr int FT = KC;                           // 16-feature tiles
    constexpr int FS = bwd_fsplit(D, KC);            // feature slices (blocks per group of atom tiles)
    constexpr int FTW = (FT + FS - 1) / FS;          // feature tiles per slice
    constexpr int FPW = 16 * FTW;                    // floats per bank row held in LDS
    constexpr int SB = bwd_slice_stride(FPW);        // LDS row stride
    constexpr int LQ = bwd_lq(D);
    constexpr int NWV = NT / 64;
    extern __shared__ __align__(16) float lds[];
    const int L = a.L;
    float* bank = lds;                               // [(D+1)*L][SB] (this block's feature slice), row b*L + l; rows D*L + l = centres
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane & 15, kq = lane >> 4;
    unsigned long long* stamps = g_rows_stamp_buffer
        ? g_rows_stamp_buffer + (((size_t)(D - 1) * 1024 + blockIdx.x) * NWV + wave) * 32 : nullptr;
    int sslot = 2;
    ROWS_STAMP(0);
    const float w_s = a.mix[0], w_c = a.mix[1], w_sum = a.mix[3];
    const float ws_n = w_s / w_sum / (float)D;
    const float ratio_c = w_c * (float)D / w_s;
    const int64_t ntiles = (a.n + 15) / 16;
    // Coefficient inputs of a tile, software pipelined one tile ahead: the loads of tile t + 1 are issued
    // before the MFMA loop of tile t and its focal ids (the address of the grad_out gather) one tile before
    // that, so the two dependent global round trips per tile overlap the matrix work instead of preceding it.
    const int slice = blockIdx.x % FS;               // (the host launches a multiple of FS blocks)
    const int ft0 = slice * FTW;                     // this block's feature tiles: ft0 .. ft0 + FTW - 1 (those < FT exist)
    const int64_t tstep = (int64_t)(gridDim.x / FS) * NWV;
    float rg[LQ];
    int ridx[LQ], rch[LQ];
    auto focal_of = [&](int64_t tile) -> int64_t {
        const int64_t n = tile * 16 + ci;
        return a.sel[n < a.n ? n : a.n - 1];
    };
    const int8_t* chp = a.chir ? a.chir : (const int8_t*)a.pair;      // always loadable; ignored when there are no signs
    auto issue = [&](int64_t tile, int64_t focal, float (&g)[LQ], int (&ix)[LQ], int (&ch)[LQ]) {
        const int64_t n = tile * 16 + ci;
        const int64_t nc = n < a.n ? n : a.n - 1;
        // Loads are unconditional on clamped addresses and masked afterwards: a load under a lane-dependent
        // branch ends its own basic block with a full wait, which serialises the round trips.
#pragma unroll
        for (int kk = 0; kk < LQ; ++kk) {
            const int l = 4 * kk + kq < L ? 4 * kk + kq : L - 1;
            g[kk] = a.gout[focal * a.gs + a.off + l];
            ix[kk] = pair_index(a.pair, (size_t)nc * L + l);
            ch[kk] = chp[(size_t)nc * L + l];
        }
    };
    int64_t tile = (int64_t)(blockIdx.x / FS) * NWV + wave;
    int64_t focal_next = 0;
    {   // unconditional (clamped past the end): a load inside a conditional block is waited for at its end
        const int64_t t0 = tile < ntiles ? tile : ntiles - 1;
        issue(t0, focal_of(t0), rg, ridx, rch);
        focal_next = focal_of(tile + tstep);
    }
    // ---- one-time: the unit kernel rows -> LDS.  Issued AFTER the first tile's coefficient loads, which then
    // complete under the copy (they were an exposed double round trip of 4-10 k cycles per wave, and at batch
    // 4096 a wave has one to three tiles); 16 loads in flight per thread.
    {
        constexpr int CPR = FPW / 4;                 // 16-byte chunks per slice row
        const int nchunks = (D + 1) * L * CPR;
        for (int base = 0; base < nchunks; base += NT * 16) {
            f32x4 tmp[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int q = base + tid + NT * k;
                const int qc = q < nchunks ? q : 0;
                const int row = qc / CPR, c = qc - row * CPR;
                const int col = ft0 * 16 + 4 * c;    // column in the full padded row; a slice may run past FP
                tmp[k] = *(const f32x4*)(a.padded + (size_t)row * FP + (col < FP ? col : 0));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int q = base + tid + NT * k;
                if (q < nchunks) {
                    const int row = q / CPR, c = q - row * CPR;
                    *(f32x4*)(bank + (size_t)row * SB + 4 * c) = tmp[k];
                }
            }
        }
    }
    __syncthreads();
    ROWS_STAMP(1);
    for (; tile < ntiles; tile += tstep) {
        const int64_t n = tile * 16 + ci;
        const bool row_ok = n < a.n;
        float c[LQ];
        int pk[LQ];
#pragma unroll
        for (int kk = 0; kk < LQ; ++kk) {
            const bool ok = row_ok && (4 * kk + kq < L);
            const float g = ok ? (a.chir ? rg[kk] * (float)rch[kk] : rg[kk]) : 0.f;
            int bits = 0;
#pragma unroll
            for (int s = 0; s < D; ++s) bits |= perm_at<D>(ridx[kk], s) << (2 * s);
            pk[kk] = bits;
            c[kk] = g * ws_n;
        }
        ROWS_STAMP(sslot);
        // next tile's inputs: in flight during this tile's MFMAs (tiles past the end read clamped rows, unused).
        constexpr bool PF = true;
        if constexpr (PF) {
            issue(tile + tstep, focal_next, rg, ridx, rch);
            focal_next = focal_of(tile + 2 * tstep);
        }
        f32x4 acc[D + 1][FTW];                       // slot 0 = focal, 1 + s = neighbour s
#pragma unroll
        for (int s = 0; s <= D; ++s)
#pragma unroll
            for (int ft = 0; ft < FTW; ++ft) acc[s][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < LQ; ++kk) {
            if (4 * kk < L) {                        // wave-uniform
                const int l = 4 * kk + kq < L ? 4 * kk + kq : L - 1;
                const float* brow = bank + (size_t)l * SB + ci;
                // all B values of this kernel group first (their LDS reads travel together), then the MFMAs
                // (a slice that runs past the last feature tile holds a copy of the row's first columns there; the product is not stored)
                float bv[D + 1][FTW];
#pragma unroll
                for (int b = 0; b <= D; ++b)
#pragma unroll
                    for (int ft = 0; ft < FTW; ++ft)
                        bv[b][ft] = brow[(size_t)b * L * SB + 16 * ft];
                {   // centre rows -> focal slot
                    const float av = c[kk] * ratio_c;
#pragma unroll
                    for (int ft = 0; ft < FTW; ++ft)
                        acc[0][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[D][ft], acc[0][ft], 0, 0, 0);
                }
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    float av[D];
#pragma unroll
                    for (int s = 0; s < D; ++s) av[s] = (((pk[kk] >> (2 * s)) & 3) == b) ? c[kk] : 0.f;
#pragma unroll
                    for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
                        for (int s = 0; s < D; ++s)
                            acc[1 + s][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[b][ft], acc[1 + s][ft], 0, 0, 0);
                }
            }
        }
        ROWS_STAMP(sslot + 1);
        // ---- contribution rows: lane holds atoms kq*4 + jj, feature 16*ft + ci
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int64_t nn = tile * 16 + kq * 4 + jj;
            if (nn < a.n) {
                float* dst = a.contrib + (size_t)(a.contrib_base + nn * (D + 1)) * a.CS + ci;
#pragma unroll
                for (int s = 0; s <= D; ++s) {
#pragma unroll
                    for (int ft = 0; ft < FTW; ++ft)
                        if (ft0 + ft < FT && 16 * (ft0 + ft) + ci < a.F)      // only the last feature tile can be partial
                            dst[(size_t)s * a.CS + 16 * (ft0 + ft)] = acc[s][ft][jj];
                }
            }
        }
        if constexpr (!PF) {
            if (tile + tstep < ntiles) issue(tile + tstep, focal_of(tile + tstep), rg, ridx, rch);
        }
        ROWS_STAMP(sslot + 2);
        sslot += 3;
    }
}

// ------------------------------------------------------------------ host ---
bool mfma_backward_supported(int d, int F, int E, int L, int64_t xs, const void* x, int64_t n_atoms) {
    if (d < 1 || d > 4 || L < 1 || E > 8) return false;
    const int FP = mfma_padded_width(F);
    if (!FP || xs % 4 != 0 || ((uintptr_t)x & 15)) return false;
    if ((uint64_t)n_atoms * (uint64_t)xs >= (1ull << 32)) return false;
    if (L > 4 * bwd_lq(d)) return false;
    return ((size_t)(d + 1) * L * FP) * 4 <= 150 * 1024;
}

template <int D, int KC>
static hipError_t launch_mfma_rows(const BwdArgs& a, int* ntheta_out, hipStream_t st) {
    // (eight waves per block were tried for degree 4 -- 112 KB of bank, one block per CU -- to give every wave a single
    // unit: at two waves per SIMD its ~300 registers spill 400 VGPRs)
    constexpr int NT = 256;
    static PerDeviceOnce attr_set;
    if (const int slot = attr_set.pending(); slot >= 0) {
        hipError_t e = hipFuncSetAttribute((const void*)kc_backward_rows_mfma<D, KC, NT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        attr_set.set(slot);
    }
    const int64_t ntiles = (a.n + 15) / 16;
    constexpr int FS = bwd_fsplit(D, KC);
    constexpr int FTW = (KC + FS - 1) / FS;
    const size_t lds_bytes = (size_t)(D + 1) * a.L * bwd_slice_stride(16 * FTW) * 4;
    int per_cu = (int)((160 * 1024 - 2048) / (lds_bytes + 256));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;                      // registers: two waves per SIMD
    constexpr int NWV = NT / 64;
    int64_t groups = 256 * per_cu / FS;              // groups of FS blocks (one per slice) that walk the atom tiles
    const int64_t need = (ntiles + NWV - 1) / NWV;
    if (groups > need) groups = need;
    if (groups < 1) groups = 1;
    int64_t blocks = groups * FS;
    if (blocks > THETA_SLAB_BLOCKS) blocks = THETA_SLAB_BLOCKS;
    kc_backward_rows_mfma<D, KC, NT><<<(int)blocks, NT, lds_bytes, st>>>(a);
    (void)ntheta_out;                                // the bank kernel sums the score-weight partials
    return hipGetLastError();
}

hipError_t launch_backward_rows_mfma(int d, const BwdArgs& a, int* ntheta_out, hipStream_t st) {
    const int KC = mfma_padded_width(a.F) / 16;
    if (KC == 2) {
        switch (d) {
            case 1: return launch_mfma_rows<1, 2>(a, ntheta_out, st);
            case 2: return launch_mfma_rows<2, 2>(a, ntheta_out, st);
            case 3: return launch_mfma_rows<3, 2>(a, ntheta_out, st);
            default: return launch_mfma_rows<4, 2>(a, ntheta_out, st);
        }
    }
    switch (d) {
        case 1: return launch_mfma_rows<1, 7>(a, ntheta_out, st);
        case 2: return launch_mfma_rows<2, 7>(a, ntheta_out, st);
        case 3: return launch_mfma_rows<3, 7>(a, ntheta_out, st);
        default: return launch_mfma_rows<4, 7>(a, ntheta_out, st);
    }
}

}  // namespace mkgnn

extern "C" int mkgnn_debug_set_rows_stamp_buffer(void* device_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mkgnn::g_rows_stamp_buffer), &device_ptr, sizeof(void*));
}
