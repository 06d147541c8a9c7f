// Kernel argument blocks and host-side launchers shared by the C ABI and the kernel files.
#pragma once
#include "kgnn_common.h"

#include <atomic>

namespace mkgnn {

struct PrepArgs {
    mkgnn_kernel_bank bank[MKGNN_MAX_DEGREE];
    float* cen[MKGNN_MAX_DEGREE];
    float* sup[MKGNN_MAX_DEGREE];
    float* edg[MKGNN_MAX_DEGREE];
    float* icen[MKGNN_MAX_DEGREE];
    float* isup[MKGNN_MAX_DEGREE];
    float* iedg[MKGNN_MAX_DEGREE];
    int8_t* chir[MKGNN_MAX_DEGREE];
    float* mix[MKGNN_MAX_DEGREE];
    float* padded[MKGNN_MAX_DEGREE];        // see BankLayout::padded
    float* edge_padded[MKGNN_MAX_DEGREE];
    int row_start[MKGNN_MAX_DEGREE + 1];   // wave-task prefix: rows of degree i are [row_start[i], row_start[i+1])
    int F, E;
};

struct FwdArgs {
    const float* x; int64_t xs; const float* inv;
    const int64_t* sel; const int64_t* nei; const float* e_nei; const float* p_focal; const float* p_nei;
    int64_t n; int F, E, L, last;
    const float* cen; const float* sup; const float* edg; const int8_t* chir; const float* mix;
    float* out; int64_t os; int off, K;
    float* pair; int8_t* chir_out;                       // saved state, atom-major: [N_d, L, 4] pair records, [N_d, L] signs
    const float* padded; const float* edge_padded;       // MFMA kernels only
    int64_t n_atoms;
};

// One launch for all degree buckets (kgnn_mfma.hip).
#ifdef MKGNN_EXP_MAX_BLOCKS                 // (occupancy experiments: make EXTRA=-DMKGNN_EXP_MAX_BLOCKS=768 ...)
constexpr int FUSED_MAX_BLOCKS = MKGNN_EXP_MAX_BLOCKS;
#else
constexpr int FUSED_MAX_BLOCKS = 512;       // 2 blocks per CU
#endif
constexpr int FUSED_MAX_GROUPS = 16;        // (degree, column part)

struct FusedDeg {
    const int64_t* sel; const int64_t* nei; const float* e_nei; const float* e_unit; const float* p_focal; const float* p_nei;
    const float* padded; const float* edge_padded; const int8_t* chir; const float* mix; const int8_t* eqflag; const int8_t* signflag;
    float* pair; int8_t* chir_out;
    int64_t n;
    int L, off;
    int nct;        // column tiles (<= 16 kernels each)
    int kpt;        // kernels per column tile
    int cs;         // column split: part cp takes the column tiles cp, cp + cs, ...
    int nloc;       // column tiles resident in a block = ceil(nct / cs)
    int ics;        // waves of a block that share one atom tile; wave w takes the resident tiles w % ics, + ics, ...
};

struct FusedFwdArgs {
    const float* x; int64_t xs; const float* inv;
    float* out; int64_t os;
    int K, F, E, last;
    int64_t n_atoms;
    int bf16;                                // node-feature dot products with bf16 operands (variant 3)
    int x_split;                             // the rows of x are pre-split (MKGNN_VARIANT_ROWS_SPLIT; kgnn_split.h)
    int FPB;                                 // row pitch (floats) of the padded bank copies (streamed kernel; = mfma_padded_width(F))
    FusedDeg deg[MKGNN_MAX_DEGREE];
    uint8_t grp_degree[FUSED_MAX_GROUPS];   // group -> degree index (0..3)
    uint8_t grp_cp[FUSED_MAX_GROUPS];       // group -> column part
    uint16_t grp_count[FUSED_MAX_GROUPS];   // blocks in the group
    uint8_t blk_group[FUSED_MAX_BLOCKS];    // block -> group
    uint16_t blk_rank[FUSED_MAX_BLOCKS];    // block -> rank inside its group
    unsigned long long* stamps;             // diagnostics (tools/stream_stamps.py): per wave {start, end, group, iterations}; usually null
};

struct BwdArgs {
    const float* x; int64_t xs; const float* inv;
    const int64_t* sel; const int64_t* nei; const float* e_nei;
    int64_t n; int F, E, L;
    const float* cen; const float* sup; const float* edg; const float* mix;
    const float* gout; int64_t gs; int off;
    const float* pair; const int8_t* chir;
    float* contrib; int64_t contrib_base;      // rows base + n*(D+1) + slot
    int CS;                                    // contrib row stride: F rounded up to 4 (16-byte rows for the gather)
    float* slab; int nchunk;                   // [nchunk, bank_floats]
    const float* padded;                       // unit bank rows, support-major, padded (LDS kernels)
    float* theta_slab;                         // [blocks][4] score-weight partials (LDS rows kernel, or the bank kernel)
    int theta_in_bank;                         // bank kernel sums the score-weight partials (MFMA rows kernel in use)
};

struct BankReduceArgs {
    const float* slab; int nchunk; int F, E, L;
    const float* theta_src; size_t theta_stride; int theta_count;   // score-weight partials
    const float* cen; const float* sup; const float* edg;
    const float* icen; const float* isup; const float* iedg;
    mkgnn_kernel_bank_grad g;
};

hipError_t launch_row_inv_norm(const float* x, int64_t stride, int64_t n, int F, float* inv, hipStream_t st);
constexpr int PREP_MANY_MAX = 4;
// arrays that spare blocks of a launch read once and throw away (mkgnn_touch_hint): 16-byte aligned interiors
constexpr int TOUCH_MAX = 16, TOUCH_BLOCKS = 512;
struct TouchArgs { const char* ptr[TOUCH_MAX]; uint32_t bytes[TOUCH_MAX]; int count; };
bool take_touch_hint(TouchArgs& out);          // this thread's pending hint, if any (kgnn_capi.hip); clears it
#ifdef __HIPCC__
// block b of nb (256 threads each): eight independent loads in flight per thread (clamped, not conditional: a load under a
// condition is issued and waited for alone); one load per trip took 12 us for 11 MB -- a chain of cold misses
__device__ __forceinline__ void touch_body(const TouchArgs& t, uint32_t b, uint32_t nb) {
    uint32_t acc = 0;
    const uint32_t stride = nb * 256u;
    for (int k = 0; k < t.count; ++k) {
        const uint4* const p = (const uint4*)t.ptr[k];
        const uint32_t n16 = t.bytes[k] >> 4;
        for (uint32_t i = b * 256u + threadIdx.x; i < n16; i += 8u * stride) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t j = i + (uint32_t)u * stride;
                v[u] = p[j < n16 ? j : n16 - 1u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    asm volatile("" :: "v"(acc));                        // (the loads stay)
}
#endif
struct PrepManyArgs { PrepArgs layer[PREP_MANY_MAX]; int task_start[PREP_MANY_MAX + 1]; int count; int prep_blocks; TouchArgs touch; };
hipError_t launch_bank_prepare_many(int count, const mkgnn_kernel_bank* banks /* [count][4] */, const WorkspaceLayout* w,
                                    char* const* ws, const int* F, int E, hipStream_t st, const TouchArgs* touch = nullptr);
// the same preparation as arguments only (tasks and blocks filled in): for a launch that carries it in blocks of its own
void build_bank_prepare_many(int count, const mkgnn_kernel_bank* banks, const WorkspaceLayout* w, char* const* ws, const int* F,
                             int E, PrepManyArgs* out);
hipError_t launch_bank_prepare_args(const PrepManyArgs& m, hipStream_t st);
// a preparation mkgnn_bank_prepare_deferred left pending on this device (kgnn_capi.hip); clears it
bool take_pending_prepare(PrepManyArgs& out);
hipError_t launch_bank_prepare(const mkgnn_kernel_bank banks[4], const WorkspaceLayout& w, char* ws, int F, int E,
                               hipStream_t st);
hipError_t launch_forward_generic(int d, const FwdArgs& a, hipStream_t st);
bool mfma_forward_supported(int d, int F, int E, int L);
int fused_group_count(int d, int F, int L);      // groups of the fused launch this degree needs (budget: FUSED_MAX_GROUPS)
hipError_t launch_forward_fused(FusedFwdArgs& a, const bool use[4], hipStream_t st);
// kgnn_fwd_stream.hip: bank in registers, atom rows streamed through LDS by DMA (the reference's shapes)
bool stream_forward_supported(int d, int F, int E, int L, int64_t n_atoms, int64_t x_stride, int64_t out_stride, const float* e_unit);
hipError_t launch_forward_stream(FusedFwdArgs& a, const bool use[4], hipStream_t st);
int stream_tiles_per_part(int d);
int stream_column_parts(int d, int L);
bool stream_rows_split_supported(int F);             // pre-split atom rows: KC <= 7 and the split-fp16 products on
bool stream_forward_bf16_supported(int F);           // the bf16 similarity variant on the streamed kernel (KC = 2 or 7)
int stream_forward_groups(const int L[4], const bool use[4]);
hipError_t launch_unit_rows8(const float* in, int64_t n_rows, int E, float* out, hipStream_t st);
hipError_t launch_backward_generic(int d, const BwdArgs& a, hipStream_t st);
struct BankReduceAllArgs { BankReduceArgs deg[4]; int blk_start[4]; };
hipError_t launch_bank_reduce_all(const BankReduceArgs r[4], hipStream_t st);   // degrees with L == 0 are skipped
// kgnn_bwd.hip: LDS-tiled backward for the model's shapes
bool lds_backward_supported(int d, int F, int E, int L, int64_t xs, const void* x);
hipError_t launch_backward_lds(int d, const BwdArgs& a, int* nchunk_out, int* ntheta_out, bool rows_too, hipStream_t st);
int bank_blocks_for(int d, int64_t n);
hipError_t launch_backward_bank_fused(const BwdArgs a4[4], const bool use[4], int nchunk_out[4], int ntheta_out[4], hipStream_t st);
// kgnn_bwd_stream.hip: streamed MFMA bank-gradient kernel (+ its coefficient pre-pass), all degrees in one launch
bool bank_stream_supported(int d, int F, int E, int L, int64_t n_atoms, int64_t x_stride, const float* e_unit);
bool bank_stream_rows_split_supported(int F);        // pre-split atom rows: KC <= 7 and the split-fp16 products on
struct BankStreamDeg {
    const int64_t* sel; const int64_t* nei; const float* e_unit;
    const float* pair; const int8_t* chir;
    const float* mix;
    float* coefq;            // [ntiles][nct][512]: g tile, idx tile
    float* slab;             // [chunks][bank_floats]
    float* theta_slab;       // [prepare blocks][4]
    int64_t n;
    int L, off, nct, kpt, cs;
    int prep_blk0, prep_blocks;      // this degree's blocks of the pre-pass
};

struct BankStreamArgs {
    const float* x; int64_t xs; const float* inv;
    const float* gout; int64_t gs;
    int F, E;
    int through_nei;         // gout is the gradient of h = propagate(out): d out[n, l] = sum over n's neighbours of gout[nei, l]
    BankStreamDeg deg[MKGNN_MAX_DEGREE];
    uint8_t grp_degree[FUSED_MAX_GROUPS];
    uint8_t grp_cp[FUSED_MAX_GROUPS];
    uint16_t grp_count[FUSED_MAX_GROUPS];
    uint8_t blk_group[FUSED_MAX_BLOCKS];
    uint16_t blk_rank[FUSED_MAX_BLOCKS];
};

// kgnn_tail.hip: the middle of the fused tail (mkgnn_tail_fused, kgnn_readout.hip) -- per chunk of whole molecules, on H-wide rows
struct TailMidArgs {
    const float* z; float* dz;              // [n_atoms, 32]: W1 sim (in), d loss / d z (out)
    const int32_t *rin, *cin, *rout, *cout;
    const int32_t *mol_ptr, *atom_mol;
    int64_t n_atoms, n_mols, n_loss;
    const float *b1, *w2, *b2, *wh, *bh, *y;
    int H, G;
    float drop_p; const int64_t* rng;
    float* emb; int64_t es;                 // [n_mols, G] or null
    float* pred;
    float* slab; int slab_stride;           // [blocks][slab_stride]: TAIL_* below
    int mg;                                 // molecules per group: tail_group_size(n_loss)
};
constexpr int TAIL_MAX_BLOCKS = 768;        // (three workgroups per CU)
// a workgroup's slab (floats): b1 [32] | W2 [32][32] | b2 [32] | wh [32] | bh | loss
constexpr int TAIL_B1 = 0, TAIL_W2 = 32, TAIL_B2 = 32 + 1024, TAIL_WH = TAIL_B2 + 32, TAIL_BH = TAIL_WH + 32, TAIL_LOSS = TAIL_BH + 1,
              TAIL_SLAB = (TAIL_LOSS + 1 + 3) / 4 * 4;
// the fused tail's deferred reduction (mkgnn_tail_args.defer_reduce): launched on `st` if one is pending on this thread
hipError_t launch_pending_tail_reduce(hipStream_t st, bool* launched = nullptr);
int tail_group_size(int64_t n_loss_mols);
int tail_middle_blocks(int64_t n_loss_mols);
hipError_t launch_tail_middle(const TailMidArgs& a, int nb, hipStream_t st);

struct BankStreamLaunch { BankStreamArgs a; int nb, prep_blocks, KC; size_t lds_bytes; int x_split; };
// block split and arguments once; then the pre-pass (coefficient records in tile order, score-weight partials) and the
// bank kernel, each on the stream the caller chooses
void plan_backward_bank_stream(const BwdArgs a4[4], const bool use[4], const float* const e_unit[4], float* const coefq[4],
                               int nchunk_out[4], int ntheta_out[4], bool through_nei, BankStreamLaunch* out);
hipError_t launch_coef_prepare(const BankStreamLaunch& p, hipStream_t st);
hipError_t launch_backward_bank_stream(const BankStreamLaunch& p, hipStream_t st);
// kgnn_bwd_rows_stream.hip: streamed MFMA rows kernel (bank in registers), all degrees in one launch
bool rows_stream_supported(int d, int F, int E, int L);
// coefq / nct: the pre-pass's records (null: the kernel gathers its coefficients itself)
hipError_t launch_backward_rows_stream(const BwdArgs a4[4], const bool use[4], float* const coefq[4], hipStream_t st);
// kgnn_bwd_mfma.hip: MFMA backward for the model's shapes
bool mfma_backward_supported(int d, int F, int E, int L, int64_t xs, const void* x, int64_t n_atoms);
hipError_t launch_backward_rows_mfma(int d, const BwdArgs& a, int* ntheta_out, hipStream_t st);
hipError_t launch_backward_gather(const float* contrib, int64_t cs, int64_t n_contrib_rows, const int32_t* rowptr,
                                  const int32_t* rows, const float* x, int64_t xs, const float* inv, int64_t n, int F,
                                  float* gx, int64_t gxs, bool allow_fast, hipStream_t st, bool x_split = false);
// kgnn_csr.hip: pipelined variants for 16-byte aligned rows of <= 256 floats; false = not applicable
bool segment_sum_blocks_supported(const float* in, int64_t is, int64_t n, int width, const float* out, int64_t os);
hipError_t launch_segment_sum_blocks(const float* in, int64_t is, const int32_t* rowptr, const int32_t* col, const int8_t* deg8,
                                     int64_t n, int width, const int32_t L[4], int mode, float* out, int64_t os,
                                     float* inv_norm, hipStream_t st);
bool try_segment_sum_aligned(const float* in, int64_t is, const int32_t* rowptr, const int32_t* col, int64_t n, int width,
                             float* out, int64_t os, float* inv_norm, hipStream_t st, hipError_t* err);
bool try_backward_gather_aligned(const float* contrib, int64_t cs, const int32_t* rowptr, const int32_t* rows, const float* x,
                                 int64_t xs, const float* inv, int64_t n, int F, float* gx, int64_t gxs, hipStream_t st,
                                 hipError_t* err, bool x_split = false);
bool try_rows_presplit(const float* x, int64_t xs, int64_t n, int width, float* inv, float* out, int64_t os, hipStream_t st, hipError_t* err);
bool try_row_inv_norm_aligned(const float* x, int64_t xs, int64_t n, int width, float* inv, hipStream_t st, hipError_t* err);
hipError_t launch_segment_sum(const float* in, int64_t is, const int32_t* rowptr, const int32_t* col, int64_t n,
                              int width, float* out, int64_t os, float* inv_norm, hipStream_t st);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: remember per device whether
// it has been set (setting it twice from two threads is harmless, skipping it on a second GPU is a failed launch).
struct PerDeviceOnce {
    std::atomic<bool> done[16];
    PerDeviceOnce() { for (auto& d : done) d.store(false); }
    // returns the device slot if the attribute still has to be set for the current device, -1 if it is set
    int pending() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;   // unknown device: set it every time
        return done[dev].load(std::memory_order_acquire) ? -1 : dev;
    }
    void set(int dev) { if (dev >= 0 && dev < 16) done[dev].store(true, std::memory_order_release); }
};

// Diagnostics of the three streamed kernels' block split (mkgnn_debug_set_grid_caps / mkgnn_debug_last_plans, kgnn_capi.hip):
// a run-time cap on the grid (0 = the kernel's default; read at every launch, so one test process can change it) -- with a
// small cap every stream owns many tiles, which is how the parity tests reach the multi-tile steady state of the
// benchmark at oracle-sized batches -- and what the last launch of each kernel was split into.
struct GridCaps { std::atomic<int> fwd{0}, rows{0}, bank{0}; };
extern GridCaps g_grid_caps;
struct PlanInfo { std::atomic<int> blocks{0}, min_iters{0}, max_iters{0}, launches{0}; };
extern PlanInfo g_last_plan[3];              // 0 forward, 1 rows gradient, 2 bank gradient
inline int grid_cap(const std::atomic<int>& cap, int dflt) {
    const int c = cap.load(std::memory_order_relaxed);
    return (c >= 8 && c <= FUSED_MAX_BLOCKS) ? c : dflt;
}
inline void note_plan(int which, int blocks, int ng, const int64_t* tiles_of, const int* count, const int* nstream_of) {
    int lo = 1 << 30, hi = 0;
    for (int g = 0; g < ng; ++g) {
        const int64_t streams = (int64_t)count[g] * nstream_of[g];
        const int it = (int)((tiles_of[g] + streams - 1) / streams);
        if (it < lo) lo = it;
        if (it > hi) hi = it;
    }
    g_last_plan[which].blocks.store(blocks); g_last_plan[which].min_iters.store(ng ? lo : 0); g_last_plan[which].max_iters.store(hi);
}

// error reporting shared by the C-ABI translation units (kgnn_capi.hip owns the thread-local message)
int api_fail(const char* fmt, ...);
int api_hip_fail(const char* what, hipError_t e);

// MKGNN_BWD_SPLIT / mkgnn_debug_set_backward_products: the streamed backward kernels' products as split fp16 (kgnn_split.h)
int bwd_split_mode();

}  // namespace mkgnn
