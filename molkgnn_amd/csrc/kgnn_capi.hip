// C ABI of libmolkgnn_hip.so (include/molkgnn_hip.h): argument checks, workspace
// carving and kernel launches.  No allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "kgnn_launch.h"
#include "kgnn_split.h"

using namespace mkgnn;

static thread_local char g_err[512] = "";

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

static int hip_fail(const char* what, hipError_t e) {
    return fail("%s: %s", what, hipGetErrorString(e));
}

namespace mkgnn {
int api_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
int api_hip_fail(const char* what, hipError_t e) { return fail("%s: %s", what, hipGetErrorString(e)); }
}  // namespace mkgnn

// The four degree buckets are independent.  At the batch sizes a molecule model sees, one bucket
// does not fill 256 CUs for long (a few atom tiles per SIMD, and prologue/tail dominate), so the
// per-degree kernels of one call run concurrently on three helper streams forked from and joined
// back into the caller's stream with events (a fork/join that hipGraph capture records as is).
// Streams and events are created once per device on first use.
struct DegreeStreams {
    hipStream_t aux[3];
    hipEvent_t fork;
    hipEvent_t join[3];
    bool ready;
    bool deferred;      // aux[0] carries bank-gradient chains nobody has joined yet (MKGNN_BACKWARD_DEFER_BANK)
    std::once_flag once;
};
static DegreeStreams g_streams[16];

// Created once per device under std::call_once (two host threads making their first call on one device do not race on
// the table).  The fork / join events are per device: the header's threading rule -- one host thread per device inside
// the backward call at a time -- is what keeps two callers from recording the same event.
static DegreeStreams* degree_streams() {
    static const bool serial = getenv("MKGNN_SERIAL") != nullptr;    // diagnostics: keep everything on one stream
    if (serial) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    DegreeStreams& p = g_streams[dev];
    std::call_once(p.once, [&p]() {
        bool ok = true;
        // MKGNN_HELPER_PRIORITY (diagnostics): "low" / "high" = the helpers at the least / greatest stream priority
        int least = 0, greatest = 0;
        static const char* env_prio = getenv("MKGNN_HELPER_PRIORITY");
        const bool ranged = env_prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        const int prio = (ranged && env_prio[0] == 'l') ? least : ((ranged && env_prio[0] == 'h') ? greatest : 0);
        for (int i = 0; i < 3 && ok; ++i) {
            ok = ((ranged && prio != 0) ? hipStreamCreateWithPriority(&p.aux[i], hipStreamNonBlocking, prio)
                                        : hipStreamCreateWithFlags(&p.aux[i], hipStreamNonBlocking)) == hipSuccess &&
                 hipEventCreateWithFlags(&p.join[i], hipEventDisableTiming) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) == hipSuccess;
        p.ready = ok;
    });
    return p.ready ? &p : nullptr;
}

// Stream on which degree slot `i` (0 = most work) runs; slot 0 stays on the caller's stream.
struct ForkJoin {
    DegreeStreams* p;
    hipStream_t main;
    bool used[3];
    // Inside a hipGraph capture: ONE helper stream.  The x-gradient chain (rows kernels of the four degrees, then
    // the gather) stays on the caller's stream, the bank-gradient chain (bank kernels, then the reduce) runs on the
    // helper; they share nothing until the join.  Replayed graphs pay ~25 us per extra branch: one branch gains
    // (1.575 -> 1.518 ms per step), two more lose (1.634 ms), one per degree lost more.  MKGNN_FORK_MODE=0 disables it.
    bool two_way = false;
    hipError_t begin(hipStream_t st, bool enable) {
        // inside a hipGraph capture the buckets stay on the one captured stream: replayed graphs ran the
        // forked branches slower than the plain chain (measured 1.41 M vs 1.59 M molecules/s)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        static const bool fork_in_graph = getenv("MKGNN_FORK_IN_GRAPH") != nullptr;   // diagnostics: re-measure that choice
        static const char* fork_mode = getenv("MKGNN_FORK_MODE");
        const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        if (enable && capturing && !(fork_mode && fork_mode[0] == '0') && !fork_in_graph) two_way = true;
        else if (enable && !fork_in_graph && capturing) enable = false;
        main = st; p = enable ? degree_streams() : nullptr;
        used[0] = used[1] = used[2] = false;
        if (!p) return hipSuccess;
        // (inside a capture the fork point is recorded where the helper is first needed: a fork here, ahead of the pre-pass,
        // would give the kernel before it a second child in the graph -- a cross-stream edge costs the caller's chain ~5 us --
        // and the helper's kernels all follow the re-fork behind the pre-pass anyway)
        return two_way ? hipSuccess : hipEventRecord(p->fork, main);
    }
    hipStream_t stream(int slot, hipError_t* e) {
        *e = hipSuccess;
        if (!p || slot == 0) return main;
        const int i = slot - 1;
        if (!used[i]) {
            if (two_way) *e = hipEventRecord(p->fork, main);
            if (*e == hipSuccess) *e = hipStreamWaitEvent(p->aux[i], p->fork, 0);
            used[i] = true;
        }
        return p->aux[i];
    }
    // order the helper after everything enqueued on the caller's stream so far (a second fork point)
    hipError_t refork(int slot) {
        if (!p || slot == 0) return hipSuccess;
        hipError_t e = hipEventRecord(p->fork, main);
        if (e != hipSuccess) return e;
        used[slot - 1] = true;
        return hipStreamWaitEvent(p->aux[slot - 1], p->fork, 0);
    }
    hipError_t end() {
        if (!p) return hipSuccess;
        for (int i = 0; i < 3; ++i) {
            if (!used[i]) continue;
            hipError_t e = hipEventRecord(p->join[i], p->aux[i]);
            if (e != hipSuccess) return e;
            e = hipStreamWaitEvent(main, p->join[i], 0);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
};

// every degree of the call that has atoms and kernels can run on the streamed rows + bank kernels (and no diagnostic
// switch turns them off): the condition of MKGNN_BACKWARD_THROUGH_NEIGHBOURS, and of taking them for shapes round 1's
// kernels do not cover
static bool streamed_pair_covers(const mkgnn_kernel_bank banks[4], const mkgnn_degree_bucket buckets[4], const float* x,
                                 int64_t x_stride, int64_t n_atoms, int F, int E) {
    static const bool no_mfma_bwd = getenv("MKGNN_NO_MFMA_BWD") != nullptr;
    static const char* env_bank_fused = getenv("MKGNN_BANK_FUSED");
    static const char* env_rows_stream = getenv("MKGNN_ROWS_STREAM");
    static const char* env_bank_stream = getenv("MKGNN_BANK_STREAM");
    if (no_mfma_bwd || (env_bank_fused && env_bank_fused[0] == '0') || (env_rows_stream && env_rows_stream[0] == '0') ||
        (env_bank_stream && env_bank_stream[0] == '0'))
        return false;
    if ((x_stride % 4) != 0 || (((uintptr_t)x) & 15) != 0) return false;        // 16-byte rows (LDS-DMA pieces)
    bool any = false, use[4];
    int Ls[4];
    for (int i = 0; i < 4; ++i) {
        const int L = banks[i].num_kernels, d = i + 1;
        Ls[i] = L;
        use[i] = buckets[i].count > 0 && L > 0;
        if (!use[i]) continue;
        any = true;
        if ((uint64_t)buckets[i].count * (uint64_t)L >= (1ull << 32)) return false;
        if (!(rows_stream_supported(d, F, E, L) && bank_stream_supported(d, F, E, L, n_atoms, x_stride, buckets[i].nei_edge_unit)))
            return false;
    }
    return any && stream_forward_groups(Ls, use) <= FUSED_MAX_GROUPS;
}

static bool fwd_stream_enabled() {              // MKGNN_FWD_STREAM=0: round 1's forward kernel (diagnostics)
    static const char* env_stream = getenv("MKGNN_FWD_STREAM");
    return !(env_stream && env_stream[0] == '0');
}

// pre-split rows (MKGNN_VARIANT_ROWS_SPLIT / MKGNN_BACKWARD_ROWS_SPLIT): the forward dispatch puts every degree with atoms and
// kernels on the streamed kernel with split-fp16 products, and the backward on the streamed pair
static bool rows_split_covered(const mkgnn_kernel_bank banks[4], const mkgnn_degree_bucket buckets[4], const float* x, int64_t x_stride,
                               int64_t out_stride, int64_t n_atoms, int F, int E) {
    static const char* env_off = getenv("MKGNN_ROWS_SPLIT");                     // MKGNN_ROWS_SPLIT=0: never (A/B, diagnostics)
    if (env_off && env_off[0] == '0') return false;
    static const char* env_pp = getenv("MKGNN_FWD_PP");
    if (env_pp && atoi(env_pp) != 0) return false;
    if (!fwd_stream_enabled() || !stream_rows_split_supported(F) || !bank_stream_rows_split_supported(F)) return false;
    if ((x_stride % 4) != 0 || (x && (((uintptr_t)x) & 15) != 0)) return false;
    if ((uint64_t)n_atoms * (uint64_t)x_stride >= (1ull << 32)) return false;
    int Ls[4];
    bool use[4], any = false;
    for (int i = 0; i < 4; ++i) {
        Ls[i] = banks[i].num_kernels;
        use[i] = buckets[i].count > 0 && Ls[i] > 0;
        if (!use[i]) continue;
        any = true;
        if ((uint64_t)buckets[i].count * (uint64_t)Ls[i] >= (1ull << 32)) return false;
        if (!stream_forward_supported(i + 1, F, E, Ls[i], n_atoms, x_stride, out_stride, buckets[i].nei_edge_unit)) return false;
    }
    if (!any || stream_forward_groups(Ls, use) > FUSED_MAX_GROUPS) return false;
    // (a dummy aligned pointer where the caller has none yet: streamed_pair_covers only looks at its alignment)
    return streamed_pair_covers(banks, buckets, x ? x : (const float*)(uintptr_t)16, x_stride, n_atoms, F, E);
}

// degree index (0..3) -> concurrency slot, most expensive bucket first (N_d * L_d * (d*d + 1))
static void degree_slots(const mkgnn_kernel_bank banks[4], const mkgnn_degree_bucket buckets[4], int slot_of[4]) {
    double cost[4];
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < 4; ++i) cost[i] = (double)buckets[i].count * banks[i].num_kernels * ((i + 1) * (i + 1) + 1);
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b)
            if (cost[order[b]] > cost[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    for (int r = 0; r < 4; ++r) slot_of[order[r]] = r;
}

namespace mkgnn {
GridCaps g_grid_caps;
PlanInfo g_last_plan[3];
}  // namespace mkgnn

// Measurement hook for bench.py (mkgnn_debug_time_backward): when enabled, a backward call keeps all its kernels on the
// caller's stream, one after the other (no helper streams: every kernel alone on the GPU), and brackets each with HIP
// events recorded on that stream: 0 coefficient pre-pass, 1 rows gradient, 2 bank gradient, 3 bank reduce, 4 gather.
// One benchmarking thread; the flag is atomic, the events are not per caller.
static std::atomic<bool> g_time_bwd{false};
static hipEvent_t g_bwd_ev[5][2];
static bool g_bwd_ev_made = false, g_bwd_ev_used[5];
struct BwdTimer {
    hipStream_t st; int k; bool on;
    BwdTimer(hipStream_t s, int which) : st(s), k(which), on(g_time_bwd.load()) {
        if (on) { (void)hipEventRecord(g_bwd_ev[k][0], st); g_bwd_ev_used[k] = true; }
    }
    ~BwdTimer() { if (on) (void)hipEventRecord(g_bwd_ev[k][1], st); }
};

extern "C" {

// Diagnostics (not part of the drop-in ABI; used by tests/ and tools/): grid caps of the streamed forward / rows-gradient /
// bank-gradient kernels, 0 = default, otherwise 8..512, effective from the next launch; and the block split of the last
// launch of each: out[3 k ..] = {blocks, fewest, most atom tiles per stream over the (degree, column part) groups}; out[9 + k] =
// how many times streamed kernel k has been launched (forward, rows gradient, bank gradient).
int mkgnn_debug_set_grid_caps(int32_t forward_blocks, int32_t rows_blocks, int32_t bank_blocks) {
    for (int32_t v : {forward_blocks, rows_blocks, bank_blocks})
        if (v != 0 && (v < 8 || v > FUSED_MAX_BLOCKS)) return fail("mkgnn_debug_set_grid_caps: %d outside 8..%d (0 = default)", v, FUSED_MAX_BLOCKS);
    g_grid_caps.fwd.store(forward_blocks); g_grid_caps.rows.store(rows_blocks); g_grid_caps.bank.store(bank_blocks);
    return 0;
}
// reads the runtime's sticky last-error away (after a failed hipGraph capture the next launch check would report it)
int mkgnn_debug_clear_error(void) { (void)hipGetLastError(); return 0; }
int mkgnn_debug_time_backward(int32_t enable) {
    if (enable && !g_bwd_ev_made) {
        for (int k = 0; k < 5; ++k)
            for (int j = 0; j < 2; ++j)
                if (hipEventCreate(&g_bwd_ev[k][j]) != hipSuccess) return fail("mkgnn_debug_time_backward: hipEventCreate failed");
        g_bwd_ev_made = true;
    }
    for (int k = 0; k < 5; ++k) g_bwd_ev_used[k] = false;
    g_time_bwd.store(enable != 0);
    return 0;
}
// out[k] = duration (ms) of kernel k of the last timed backward call (see BwdTimer), -1 where it did not run
int mkgnn_debug_last_backward_ms(float out[5]) {
    if (!out || !g_bwd_ev_made) return fail("mkgnn_debug_last_backward_ms: timing was never enabled");
    for (int k = 0; k < 5; ++k) {
        out[k] = -1.f;
        if (!g_bwd_ev_used[k]) continue;
        if (hipEventSynchronize(g_bwd_ev[k][1]) != hipSuccess) continue;
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, g_bwd_ev[k][0], g_bwd_ev[k][1]) == hipSuccess) out[k] = ms;
    }
    return 0;
}
int mkgnn_debug_last_plans(int32_t out[12]) {
    if (!out) return fail("mkgnn_debug_last_plans: null pointer");
    for (int k = 0; k < 3; ++k) {
        out[3 * k] = g_last_plan[k].blocks.load(); out[3 * k + 1] = g_last_plan[k].min_iters.load(); out[3 * k + 2] = g_last_plan[k].max_iters.load();
        out[9 + k] = g_last_plan[k].launches.load();        // launches of the streamed kernel since the library was loaded
    }
    return 0;
}

// The power-of-two scales of the split-fp16 products (kgnn_split.h), evaluated on the HOST -- the same inline functions the kernels
// compile, plain exponent-field arithmetic -- for the exhaustive property test of tests/test_host_cpu.py (no GPU needed):
// out = bits of { split_scale_for<10>(amax), split_unscale_of<12>(that)   [the rows kernel's pair],
//                 split_scale_for_exponent<18>(exponent of amax), split_unscale_of<0>(that)   [the bank kernel's],
//                 split_row_scale_of(amax), split_row_inv(amax)   [pre-split rows / the forward's row scale, amax read as 1 / |x|] }
int mkgnn_debug_split_scales(uint32_t amax_bits, uint32_t out[6]) {
    if (!out) return fail("mkgnn_debug_split_scales: null pointer");
    const float amax = split_bits_to_float(amax_bits);
    const float s_rows = split_scale_for<10>(amax), u_rows = split_unscale_of<12>(s_rows);
    const float s_bank = split_scale_for_exponent<18>((int)((amax_bits >> 23) & 0xffu)), u_bank = split_unscale_of<0>(s_bank);
    out[0] = split_float_to_bits(s_rows); out[1] = split_float_to_bits(u_rows);
    out[2] = split_float_to_bits(s_bank); out[3] = split_float_to_bits(u_bank);
    out[4] = split_float_to_bits(split_row_scale_of(amax)); out[5] = split_float_to_bits(split_row_inv(amax));
    return 0;
}

int mkgnn_abi_version(void) { return MKGNN_ABI_VERSION; }

const char* mkgnn_last_error(void) { return g_err; }

int mkgnn_row_inv_norm(const float* x, int64_t x_stride, int64_t n_rows, int32_t F, float* inv_norm, void* stream) {
    if (n_rows < 0 || F <= 0 || x_stride < F) return fail("mkgnn_row_inv_norm: bad shape n=%lld F=%d stride=%lld",
                                                         (long long)n_rows, F, (long long)x_stride);
    if (n_rows && (!x || !inv_norm)) return fail("mkgnn_row_inv_norm: null pointer");
    hipError_t e = launch_row_inv_norm(x, x_stride, n_rows, F, inv_norm, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail("mkgnn_row_inv_norm", e);
}

int mkgnn_rows_presplit(const float* x, int64_t x_stride, int64_t n_rows, int32_t F, float* inv_norm, float* out, int64_t out_stride,
                        void* stream) {
    if (n_rows < 0 || F <= 0 || x_stride < F || out_stride < F) return fail("mkgnn_rows_presplit: bad shape");
    if (n_rows == 0) return 0;
    if (!x || !inv_norm || !out) return fail("mkgnn_rows_presplit: null pointer");
    hipError_t e = hipSuccess;
    if (!try_rows_presplit(x, x_stride, n_rows, F, inv_norm, out, out_stride, (hipStream_t)stream, &e))
        return fail("mkgnn_rows_presplit: rows of at most 256 floats, 16-byte aligned (strides multiples of 4 floats)");
    return e == hipSuccess ? 0 : hip_fail("mkgnn_rows_presplit", e);
}

int mkgnn_unit_rows8(const float* in, int64_t n_rows, int32_t E, float* out, void* stream) {
    if (n_rows < 0 || E < 1 || E > 8) return fail("mkgnn_unit_rows8: %lld rows of width %d (1..8)", (long long)n_rows, E);
    if (n_rows && (!in || !out)) return fail("mkgnn_unit_rows8: null pointer");
    hipError_t e = launch_unit_rows8(in, n_rows, E, out, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail("mkgnn_unit_rows8", e);
}

size_t mkgnn_workspace_bytes(const int32_t num_kernels[MKGNN_MAX_DEGREE], int32_t F, int32_t E, int64_t n_atoms,
                             int64_t n_edges) {
    return make_layout(num_kernels, F, E, n_atoms, n_edges).total;
}

static int check_common(const char* who, const mkgnn_kernel_bank banks[4], const mkgnn_degree_bucket buckets[4],
                        const float* x, int64_t x_stride, const float* inv_norm, int64_t n_atoms, int32_t F, int32_t E,
                        int32_t need_coordinates, int64_t* n_edges_out) {
    if (!banks || !buckets) return fail("%s: banks/buckets is null", who);
    if (F <= 0 || F > 256) return fail("%s: node attribute width F=%d outside 1..256", who, F);
    if (E <= 0 || E > 64) return fail("%s: edge attribute width E=%d outside 1..64", who, E);
    if (n_atoms < 0 || x_stride < F) return fail("%s: bad x shape", who);
    if (n_atoms && (!x || !inv_norm)) return fail("%s: x/inv_norm is null", who);
    int64_t n_edges = 0, n_focal = 0;
    for (int i = 0; i < 4; ++i) {
        const mkgnn_kernel_bank& b = banks[i];
        const mkgnn_degree_bucket& k = buckets[i];
        if (b.num_kernels < 0 || b.num_kernels > 4096) return fail("%s: degree %d has %d kernels", who, i + 1, b.num_kernels);
        if (k.count < 0) return fail("%s: degree %d bucket count %lld", who, i + 1, (long long)k.count);
        if (k.count > 0) {
            // num_kernels == 0 with atoms present: the degree is skipped here (its rows are left
            // untouched); the host raises the reference's exception (kernels.py:717-721) when
            // neither a fixed nor a trainable bank exists for such a degree.
            if (!k.selected_index || !k.nei_index || !k.nei_edge_attr)
                return fail("%s: degree %d bucket has null index/edge tensors", who, i + 1);
            if (i == 3 && need_coordinates && (!k.p_focal || !k.nei_p || !b.p_support))
                return fail("%s: degree-4 coordinates are required in the last layer (chirality)", who);
        }
        if (b.num_kernels > 0 && (!b.x_center || !b.x_support || !b.edge_attr_support || !b.support_attr_sc_weight ||
                                  !b.center_attr_sc_weight || !b.edge_attr_support_sc_weight))
            return fail("%s: degree %d bank has null parameters", who, i + 1);
        n_edges += k.count * (i + 1);
        n_focal += k.count;
    }
    if (n_focal > n_atoms) return fail("%s: buckets hold %lld atoms, batch has %lld", who, (long long)n_focal, (long long)n_atoms);
    *n_edges_out = n_edges;
    return 0;
}

static thread_local TouchArgs g_touch_hint;           // (count == 0: none pending)
extern "C++" {
namespace mkgnn {
bool take_touch_hint(TouchArgs& out) {
    if (g_touch_hint.count <= 0) return false;
    out = g_touch_hint;
    g_touch_hint.count = 0;
    return true;
}
}  // namespace mkgnn
}

int mkgnn_touch_hint(const void* const* arrays, const size_t* bytes, int32_t count) {
    if (count <= 0 || !arrays || !bytes) {
        const int had = g_touch_hint.count > 0 ? 1 : 0;
        g_touch_hint.count = 0;
        return had;
    }
    TouchArgs ta{};
    for (int k = 0; k < count && ta.count < TOUCH_MAX; ++k) {      // the 16-byte aligned interior of every array (at most 4 GiB - 16 of it)
        if (!arrays[k] || bytes[k] < 32) continue;
        const uintptr_t lo = ((uintptr_t)arrays[k] + 15) & ~(uintptr_t)15, hi = ((uintptr_t)arrays[k] + bytes[k]) & ~(uintptr_t)15;
        if (hi <= lo) continue;
        const size_t n = hi - lo;
        ta.ptr[ta.count] = (const char*)lo;
        ta.bytes[ta.count] = (uint32_t)(n > 0xfffffff0u ? 0xfffffff0u : n);
        ++ta.count;
    }
    g_touch_hint = ta;
    return 0;
}

// a preparation left pending by mkgnn_bank_prepare_deferred: per device (armed and taken by the forward's thread; the mutex keeps
// the slot whole)
struct PendingPrepare { PrepManyArgs m; bool armed; };
static PendingPrepare g_pending_prepare[16];
static std::mutex g_pending_prepare_mutex;
static PendingPrepare* pending_prepare_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    return &g_pending_prepare[dev];
}
extern "C++" {
namespace mkgnn {
bool take_pending_prepare(PrepManyArgs& out) {
    std::lock_guard<std::mutex> lock(g_pending_prepare_mutex);
    PendingPrepare* p = pending_prepare_slot();
    if (!p->armed) return false;
    out = p->m;
    p->armed = false;
    return true;
}
}  // namespace mkgnn
}

static int bank_prepare_impl(const char* who, int32_t count, const mkgnn_kernel_bank* banks, const int32_t* F, int32_t E,
                             void* const* workspaces, const size_t* workspace_bytes, bool defer, void* stream);

int mkgnn_bank_prepare_deferred(int32_t count, const mkgnn_kernel_bank* banks, const int32_t* F, int32_t E, void* const* workspaces,
                                const size_t* workspace_bytes) {
    return bank_prepare_impl("mkgnn_bank_prepare_deferred", count, banks, F, E, workspaces, workspace_bytes, true, nullptr);
}

int mkgnn_bank_prepare_flush(void* stream) {
    PrepManyArgs m;
    if (!take_pending_prepare(m)) return 0;
    (void)take_touch_hint(m.touch);
    hipError_t e = launch_bank_prepare_args(m, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail("mkgnn_bank_prepare_flush", e);
}

int mkgnn_bank_prepare_withdraw(void) {
    PrepManyArgs m;
    return take_pending_prepare(m) ? 1 : 0;
}

int mkgnn_bank_prepare(int32_t count, const mkgnn_kernel_bank* banks, const int32_t* F, int32_t E, void* const* workspaces,
                       const size_t* workspace_bytes, void* stream) {
    if (int rc = mkgnn_bank_prepare_flush(stream)) return rc;        // (one left pending and never taken: first, in order)
    return bank_prepare_impl("mkgnn_bank_prepare", count, banks, F, E, workspaces, workspace_bytes, false, stream);
}

static int bank_prepare_impl(const char* who, int32_t count, const mkgnn_kernel_bank* banks, const int32_t* F, int32_t E,
                             void* const* workspaces, const size_t* workspace_bytes, bool defer, void* stream) {
    if (count < 0 || count > PREP_MANY_MAX) return fail("%s: %d calls (0..%d per launch)", who, count, PREP_MANY_MAX);
    TouchArgs ta{};
    if (!defer) (void)take_touch_hint(ta);
    if (count == 0 && ta.count == 0) return 0;
    if (count > 0 && (!banks || !F || !workspaces || !workspace_bytes || E <= 0)) return fail("%s: bad arguments", who);
    WorkspaceLayout w[PREP_MANY_MAX];
    char* ws[PREP_MANY_MAX];
    int Fs[PREP_MANY_MAX];
    for (int k = 0; k < count; ++k) {
        int32_t L[4];
        for (int i = 0; i < 4; ++i) {
            const mkgnn_kernel_bank& b = banks[4 * k + i];
            L[i] = b.num_kernels;
            if (L[i] < 0) return fail("%s: call %d degree %d has %d kernels", who, k, i + 1, L[i]);
            if (L[i] > 0 && (!b.x_center || !b.x_support || !b.edge_attr_support || !b.support_attr_sc_weight ||
                             !b.center_attr_sc_weight || !b.edge_attr_support_sc_weight))
                return fail("%s: call %d degree %d bank has null parameters", who, k, i + 1);
        }
        if (F[k] <= 0) return fail("%s: call %d has F=%d", who, k, F[k]);
        w[k] = make_layout(L, F[k], E, 0, 0);
        if (!workspaces[k] || workspace_bytes[k] < w[k].bank[3].end)
            return fail("%s: call %d: workspace of %zu bytes, the banks need %zu", who, k, workspace_bytes[k], w[k].bank[3].end);
        ws[k] = (char*)workspaces[k];
        Fs[k] = F[k];
    }
    if (defer) {
        if (count > PREP_MANY_MAX) return fail("%s: at most %d calls", who, PREP_MANY_MAX);
        std::lock_guard<std::mutex> lock(g_pending_prepare_mutex);
        PendingPrepare* p = pending_prepare_slot();
        if (p->armed) return fail("%s: a deferred preparation is already pending on this device (mkgnn_bank_prepare_flush)", who);
        build_bank_prepare_many(count, banks, w, ws, Fs, E, &p->m);
        p->armed = p->m.prep_blocks > 0;
        return 0;
    }
    hipError_t e = launch_bank_prepare_many(count, banks, w, ws, Fs, E, (hipStream_t)stream, ta.count ? &ta : nullptr);
    return e == hipSuccess ? 0 : hip_fail(who, e);
}

int mkgnn_kernelsetconv_forward(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE],
                                const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], const float* x, int64_t x_stride,
                                const float* inv_norm, int64_t n_atoms, int32_t F, int32_t E, int32_t is_last_layer,
                                float* out, int64_t out_stride, const mkgnn_saved saved[MKGNN_MAX_DEGREE],
                                void* workspace, size_t workspace_bytes, int32_t variant, void* stream) {
    const char* who = "mkgnn_kernelsetconv_forward";
    int64_t n_edges = 0;
    if (int rc = check_common(who, banks, buckets, x, x_stride, inv_norm, n_atoms, F, E, is_last_layer, &n_edges)) return rc;
    int32_t L[4];
    int K = 0;
    for (int i = 0; i < 4; ++i) { L[i] = banks[i].num_kernels; K += L[i]; }
    if (out_stride < K || (n_atoms && !out)) return fail("%s: bad out (stride %lld, K %d)", who, (long long)out_stride, K);
    WorkspaceLayout w = make_layout(L, F, E, n_atoms, n_edges);
    if (workspace_bytes < w.fwd_end || !workspace)
        return fail("%s: workspace of %zu bytes, need %zu", who, workspace_bytes, w.fwd_end);
    const bool block_rows_only = (variant & MKGNN_VARIANT_BLOCK_ROWS) != 0;     // the caller reads only each atom's own block
    const bool bank_prepared = (variant & MKGNN_VARIANT_BANK_PREPARED) != 0;    // mkgnn_bank_prepare has filled the workspace's head
    const bool rows_split = (variant & MKGNN_VARIANT_ROWS_SPLIT) != 0;          // x is pre-split (kgnn_split.h)
    variant &= ~(MKGNN_VARIANT_BLOCK_ROWS | MKGNN_VARIANT_BANK_PREPARED | MKGNN_VARIANT_ROWS_SPLIT);
    if (variant < 0 || variant > 3) return fail("%s: variant %d", who, variant);
    if (rows_split && (variant == 1 || variant == 3 || !rows_split_covered(banks, buckets, x, x_stride, out_stride, n_atoms, F, E)))
        return fail("%s: MKGNN_VARIANT_ROWS_SPLIT needs the streamed kernels with split-fp16 products for every degree "
                    "(mkgnn_rows_split_supported(..) tells)", who);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    hipError_t e = bank_prepared ? hipSuccess : launch_bank_prepare(banks, w, ws, F, E, st);
    if (e != hipSuccess) return hip_fail("bank_prepare", e);
    // every atom's row is zero outside its own degree block (kernels.py:674-675, 725-727): one
    // streaming memset, the degree kernels then write only their column blocks
    // (alignment padding up to the next multiple of four columns is zeroed with it when the stride holds it)
    if (n_atoms > 0 && K > 0 && !block_rows_only) {
        const int64_t K4 = (K + 3) / 4 * 4;
        if (K4 == out_stride) e = hipMemsetAsync(out, 0, (size_t)n_atoms * out_stride * 4, st);   // contiguous: one fill kernel, not two
        else e = hipMemset2DAsync(out, (size_t)out_stride * 4, 0, (size_t)(K4 <= out_stride ? K4 : K) * 4, (size_t)n_atoms, st);
        if (e != hipSuccess) return hip_fail("output memset", e);
    }
    // the fused MFMA launch takes every degree whose shape it covers; the rest run on the generic kernels
    const bool aligned = (x_stride % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                         ((uint64_t)n_atoms * (uint64_t)x_stride < (1ull << 32));
    FusedFwdArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.x = x; fa.xs = x_stride; fa.inv = inv_norm; fa.out = out; fa.os = out_stride;
    fa.K = K; fa.F = F; fa.E = E; fa.last = is_last_layer ? 1 : 0; fa.n_atoms = n_atoms;
    fa.bf16 = variant == 3 ? 1 : 0;
    fa.x_split = rows_split ? 1 : 0;
    bool use[4] = {false, false, false, false};
    bool any_fused = false;
    // which degrees ride in the fused launch: every covered shape, as long as their (degree, column part) groups fit
    // the launch's group table -- very wide banks (many 16-kernel column tiles) are demoted to the generic kernels,
    // largest first
    bool fuse[4];
    int groups[4], total_groups = 0;
    int only_groups[4] = {0, 0, 0, 0}, total_only = 0;      // (degree, column part) groups of the degrees ONLY the streamed kernel covers
    for (int i = 0; i < 4; ++i) {
        // (the fused kernel indexes the saved planes [3, N_d, L] with 32-bit offsets)
        const bool small = (uint64_t)buckets[i].count * (uint64_t)L[i] < (1ull << 32);     // (pair offsets are 32-bit)
        // (rows wider than round 1's kernels take, F > 112: only the streamed kernel, so only the exact-fp32 variants)
        const bool old_ok = mfma_forward_supported(i + 1, F, E, L[i]);
        const bool stream_only = !old_ok && variant != 3 && fwd_stream_enabled() &&
                                 stream_forward_supported(i + 1, F, E, L[i], n_atoms, x_stride, out_stride, buckets[i].nei_edge_unit);
        fuse[i] = buckets[i].count > 0 && L[i] > 0 && variant != 1 && aligned && small && (old_ok || stream_only);
        groups[i] = (fuse[i] && old_ok) ? fused_group_count(i + 1, F, L[i]) : 0;
        total_groups += groups[i];
        only_groups[i] = (fuse[i] && !old_ok) ? stream_column_parts(i + 1, L[i]) : 0;
        total_only += only_groups[i];
    }
    // a degree only the streamed kernel covers cannot be handed to the LDS-bank kernel when the streamed launch's group table
    // overflows (launch_forward_fused demotes covered degrees only): more such groups than the table holds go to the generic kernels
    while (total_only > FUSED_MAX_GROUPS) {
        int big = 0;
        for (int i = 1; i < 4; ++i) if (only_groups[i] > only_groups[big]) big = i;
        if (variant >= 2)
            return fail("%s: the banks need %d column groups, the streamed launch holds %d", who, total_only, FUSED_MAX_GROUPS);
        total_only -= only_groups[big]; only_groups[big] = 0; fuse[big] = false;
    }
    while (total_groups > FUSED_MAX_GROUPS) {
        int big = 0;
        for (int i = 1; i < 4; ++i) if (groups[i] > groups[big]) big = i;
        if (variant >= 2)
            return fail("%s: the banks need %d column groups, the fused launch holds %d", who, total_groups, FUSED_MAX_GROUPS);
        total_groups -= groups[big]; groups[big] = 0; fuse[big] = false;
    }
    int off = 0;
    for (int i = 0; i < 4; ++i) {
        const int d = i + 1;
        FwdArgs a;
        a.x = x; a.xs = x_stride; a.inv = inv_norm;
        a.sel = buckets[i].selected_index; a.nei = buckets[i].nei_index; a.e_nei = buckets[i].nei_edge_attr;
        a.p_focal = buckets[i].p_focal; a.p_nei = buckets[i].nei_p;
        a.n = buckets[i].count; a.F = F; a.E = E; a.L = L[i]; a.last = is_last_layer ? 1 : 0;
        a.cen = (const float*)(ws + w.bank[i].cen); a.sup = (const float*)(ws + w.bank[i].sup);
        a.edg = (const float*)(ws + w.bank[i].edg); a.chir = (const int8_t*)(ws + w.bank[i].chir);
        a.mix = (const float*)(ws + w.bank[i].mix);
        a.padded = (const float*)(ws + w.bank[i].padded); a.edge_padded = (const float*)(ws + w.bank[i].edge_padded);
        a.n_atoms = n_atoms;
        a.out = out; a.os = out_stride; a.off = off; a.K = K;
        a.pair = saved ? saved[i].pair_state : nullptr;
        a.chir_out = (saved && d == 4 && is_last_layer) ? saved[i].chirality : nullptr;
        off += L[i];
        if (a.n == 0 || a.L == 0) continue;
        const bool can_fuse = fuse[i];
        if (variant >= 2 && !can_fuse)
            return fail("%s: MFMA variant does not cover degree %d with F=%d E=%d L=%d stride=%lld", who, d, F, E, L[i],
                        (long long)x_stride);
        if (can_fuse) {
            FusedDeg& g = fa.deg[i];
            g.sel = a.sel; g.nei = a.nei; g.e_nei = a.e_nei; g.e_unit = buckets[i].nei_edge_unit; g.p_focal = a.p_focal; g.p_nei = a.p_nei;
            g.padded = a.padded; g.edge_padded = a.edge_padded; g.chir = a.chir; g.mix = a.mix;
            g.eqflag = (const int8_t*)(ws + w.eqflag);
            g.signflag = (const int8_t*)(ws + w.signflag);
            g.pair = a.pair; g.chir_out = a.chir_out;
            g.n = a.n; g.L = a.L; g.off = a.off;
            use[i] = true;
            any_fused = true;
        } else {
            e = launch_forward_generic(d, a, st);
            if (e != hipSuccess) return hip_fail("kernelconv forward launch", e);
        }
    }
    if (any_fused) {
        e = launch_forward_fused(fa, use, st);
        if (e != hipSuccess) return hip_fail("fused kernelconv forward launch", e);
    }
    return 0;
}

int mkgnn_kernelsetconv_backward(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE],
                                 const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], const float* x, int64_t x_stride,
                                 const float* inv_norm, int64_t n_atoms, int32_t F, int32_t E, int32_t is_last_layer,
                                 const float* grad_out, int64_t grad_out_stride,
                                 const mkgnn_saved saved[MKGNN_MAX_DEGREE], const int32_t* scatter_rowptr,
                                 const int32_t* scatter_rows, float* grad_x, int64_t grad_x_stride,
                                 const mkgnn_kernel_bank_grad grads[MKGNN_MAX_DEGREE], void* workspace,
                                 size_t workspace_bytes, int32_t workspace_from_forward, int32_t variant, void* stream) {
    const char* who = "mkgnn_kernelsetconv_backward";
    const bool defer_bank = (variant & MKGNN_BACKWARD_DEFER_BANK) != 0;
    const bool through_nei = (variant & MKGNN_BACKWARD_THROUGH_NEIGHBOURS) != 0;
    const bool rows_split = (variant & MKGNN_BACKWARD_ROWS_SPLIT) != 0;         // x is pre-split (kgnn_split.h)
    variant &= ~(MKGNN_BACKWARD_DEFER_BANK | MKGNN_BACKWARD_THROUGH_NEIGHBOURS | MKGNN_BACKWARD_ROWS_SPLIT);
    if (variant < 0 || variant > 2) return fail("%s: variant %d (0 = automatic, 1 = generic kernels, 2 = fast kernels)", who, variant);
    const bool force_generic = variant == 1, force_fast = variant == 2;
    int64_t n_edges = 0;
    if (int rc = check_common(who, banks, buckets, x, x_stride, inv_norm, n_atoms, F, E, 0, &n_edges)) return rc;
    if (!saved || !grads) return fail("%s: saved/grads is null", who);
    int32_t L[4];
    int K = 0;
    for (int i = 0; i < 4; ++i) { L[i] = banks[i].num_kernels; K += L[i]; }
    if (grad_out_stride < K || (n_atoms && !grad_out)) return fail("%s: bad grad_out", who);
    if (grad_x && grad_x_stride < F) return fail("%s: bad grad_x stride", who);
    if (grad_x && n_atoms && (!scatter_rowptr || !scatter_rows)) return fail("%s: scatter CSR is null", who);
    WorkspaceLayout w = make_layout(L, F, E, n_atoms, n_edges);
    if (workspace_bytes < w.total || !workspace) return fail("%s: workspace of %zu bytes, need %zu", who, workspace_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    hipError_t e = hipSuccess;
    if (!workspace_from_forward) {                   // else the normalised bank of the forward call is still there
        e = launch_bank_prepare(banks, w, ws, F, E, st);
        if (e != hipSuccess) return hip_fail("bank_prepare", e);
    }
    int slot_of[4];
    degree_slots(banks, buckets, slot_of);
    ForkJoin fj;
    e = fj.begin(st, !g_time_bwd.load());            // (timed for bench.py: everything on the caller's stream)
    if (e != hipSuccess) return hip_fail("stream fork", e);
    int off = 0;
    int64_t base = 0;
    BankReduceArgs reduce[4];
    bool bank_on_main = false;                       // some bank gradient was computed on the caller's stream
    int off_of[4]; int64_t base_of[4];
    for (int i = 0; i < 4; ++i) { off_of[i] = off; base_of[i] = base; off += L[i]; base += buckets[i].count * (i + 2); }
    // The bank gradients of all four degrees in one launch when every degree runs the MFMA rows + LDS bank pair: the
    // blocks of the next degree start as soon as a CU is free.  Neutral at batch 4096 (the replayed graph is bound by the
    // combined work of its two chains), 4-10 % of a step at batches <= 2048, where four launches of a few latency-bound
    // blocks each ran one after the other.
    static const bool no_mfma_bwd = getenv("MKGNN_NO_MFMA_BWD") != nullptr;     // diagnostics: A/B against the LDS rows kernel
    static const char* env_bank_fused = getenv("MKGNN_BANK_FUSED");             // diagnostics: "0" = one launch per degree
    static const char* env_rows_stream = getenv("MKGNN_ROWS_STREAM");
    static const char* env_bank_stream = getenv("MKGNN_BANK_STREAM");
    // the streamed pair (rows + bank kernels on the pre-pass's records) covers every degree of this call: then it runs
    // whatever round 1's kernels would have said about the shapes
    const bool stream_all = streamed_pair_covers(banks, buckets, x, x_stride, n_atoms, F, E) && !force_generic;
    bool fuse_bank = !(env_bank_fused && env_bank_fused[0] == '0') && !no_mfma_bwd && !force_generic;
    for (int i = 0; i < 4 && fuse_bank && !stream_all; ++i)
        if (buckets[i].count > 0 && L[i] > 0 &&
            !(lds_backward_supported(i + 1, F, E, L[i], x_stride, x) && mfma_backward_supported(i + 1, F, E, L[i], x_stride, x, n_atoms)))
            fuse_bank = false;
    // the x-gradient rows of all degrees in one streamed launch when every degree's shape is covered (kgnn_bwd_rows_stream.hip);
    // MKGNN_ROWS_STREAM=0: one kc_backward_rows_mfma launch per degree (diagnostics)
    // (without grad_x -- a layer whose input carries no gradient -- nobody reads contribution rows: with the bank gradients
    // fused, which then sum the score-weight partials themselves, no rows kernel is launched at all)
    bool rows_streamed = fuse_bank && !(env_rows_stream && env_rows_stream[0] == '0');
    for (int i = 0; i < 4 && rows_streamed; ++i)
        if (buckets[i].count > 0 && L[i] > 0 && !rows_stream_supported(i + 1, F, E, L[i])) rows_streamed = false;
    BwdArgs bank_a[4];
    bool bank_use[4] = {false, false, false, false};
    for (int i = 0; i < 4; ++i) {                    // (the launch order of the degrees makes no measurable difference)
        const int d = i + 1;
        const int off = off_of[i];
        const int64_t base = base_of[i];
        hipStream_t dst = fj.stream(fj.two_way ? 0 : slot_of[i], &e);
        if (e != hipSuccess) return hip_fail("stream fork", e);
        if (buckets[i].count > 0 && L[i] > 0 && !saved[i].pair_state)
            return fail("%s: degree %d has no saved forward state", who, d);
        BwdArgs a;
        a.x = x; a.xs = x_stride; a.inv = inv_norm;
        a.sel = buckets[i].selected_index; a.nei = buckets[i].nei_index; a.e_nei = buckets[i].nei_edge_attr;
        a.n = buckets[i].count; a.F = F; a.E = E; a.L = L[i];
        a.cen = (const float*)(ws + w.bank[i].cen); a.sup = (const float*)(ws + w.bank[i].sup);
        a.edg = (const float*)(ws + w.bank[i].edg); a.mix = (const float*)(ws + w.bank[i].mix);
        a.gout = grad_out; a.gs = grad_out_stride; a.off = off;
        a.pair = saved[i].pair_state;
        a.chir = (d == 4 && is_last_layer) ? saved[i].chirality : nullptr;
        if (d == 4 && is_last_layer && buckets[i].count > 0 && !a.chir)
            return fail("%s: degree-4 chirality signs were not saved", who);
        a.contrib = (float*)(ws + w.contrib); a.contrib_base = base; a.CS = (F + 3) / 4 * 4;
        a.slab = (float*)(ws + w.slab_off[i]);
        a.padded = (const float*)(ws + w.bank[i].padded);
        int64_t nc = a.n < BWD_BANK_BLOCKS ? (a.n > 0 ? a.n : 1) : BWD_BANK_BLOCKS;
        a.nchunk = (int)nc;
        a.theta_slab = (float*)(ws + w.theta_off[i]);
        int nchunk = a.n > 0 ? a.nchunk : 0;
        int ntheta = -1;
        if (a.n > 0 && L[i] == 0 && grad_x) {
            // atoms of this degree but no kernels for it in this set (a fixed / trainable split, kernels.py:699-720): the
            // scatter CSR still points at their contribution rows, which nobody writes -- they contribute zero
            e = hipMemsetAsync(a.contrib + (size_t)base * a.CS, 0, (size_t)a.n * (d + 1) * a.CS * sizeof(float), dst);
            if (e != hipSuccess) return hip_fail("contribution rows memset", e);
        }
        if (a.n > 0 && L[i] > 0) {
            const bool fast_ok = stream_all || lds_backward_supported(d, F, E, L[i], x_stride, x);
            if (force_fast && !stream_all && !(fast_ok && mfma_backward_supported(d, F, E, L[i], x_stride, x, n_atoms)))
                return fail("%s: the fast kernels do not cover degree %d with F=%d E=%d L=%d stride=%lld", who, d, F, E, L[i],
                            (long long)x_stride);
            if (fast_ok && !force_generic) {
                const bool rows_mfma = stream_all || (!no_mfma_bwd && mfma_backward_supported(d, F, E, L[i], x_stride, x, n_atoms));
                hipStream_t st_rows = dst, st_bank = dst;
                if (fj.two_way && !rows_mfma) bank_on_main = true;
                if (fj.two_way && rows_mfma) {
                    st_rows = st;
                    if (!fuse_bank) {                        // (fused: the bank kernels are launched below, behind the pre-pass's re-fork)
                        st_bank = fj.stream(1, &e);
                        if (e != hipSuccess) return hip_fail("stream fork", e);
                    }
                }
                if (rows_mfma && !rows_streamed && (grad_x || !fuse_bank)) {
                    e = launch_backward_rows_mfma(d, a, &ntheta, st_rows);
                    if (e != hipSuccess) return hip_fail("kernelconv backward launch", e);
                }
                if (fuse_bank) { bank_a[i] = a; bank_use[i] = true; }       // launched below, all degrees together
                else e = launch_backward_lds(d, a, &nchunk, &ntheta, !rows_mfma, st_bank);
            }
            else { e = launch_backward_generic(d, a, dst); bank_on_main = true; }
            if (e != hipSuccess) return hip_fail("kernelconv backward launch", e);
        }
        BankReduceArgs& r = reduce[i];
        r.slab = a.slab; r.nchunk = nchunk; r.F = F; r.E = E; r.L = L[i];
        if (ntheta >= 0) { r.theta_src = a.theta_slab; r.theta_stride = 4; r.theta_count = ntheta; }
        else {
            r.theta_src = a.slab + (size_t)L[i] * F + (size_t)L[i] * d * F + (size_t)L[i] * d * E;
            r.theta_stride = bank_floats(d, L[i], F, E); r.theta_count = nchunk;
        }
        r.cen = a.cen; r.sup = a.sup; r.edg = a.edg;
        r.icen = (const float*)(ws + w.bank[i].icen); r.isup = (const float*)(ws + w.bank[i].isup);
        r.iedg = (const float*)(ws + w.bank[i].iedg);
        r.g = grads[i];
    }
    const bool any_bank = bank_use[0] || bank_use[1] || bank_use[2] || bank_use[3];
    // the reference's bank shapes: the streamed MFMA kernel (kgnn_bwd_stream.hip); anything else: the LDS / VALU one.
    // MKGNN_BANK_STREAM=0: A/B switch (diagnostics)
    bool streamed = any_bank && !(env_bank_stream && env_bank_stream[0] == '0');
    {   // (the streamed launches hold FUSED_MAX_GROUPS (degree, column part) groups)
        int Lu[4];
        for (int i = 0; i < 4; ++i) Lu[i] = L[i];
        if (stream_forward_groups(Lu, bank_use) > FUSED_MAX_GROUPS) streamed = false;
    }
    const float* e_unit4[4]; float* coefq4[4];
    {
        size_t coef_off = w.coefq_off[0];
        for (int i = 0; i < 4; ++i) {
            e_unit4[i] = buckets[i].nei_edge_unit; coefq4[i] = nullptr;
            if (!bank_use[i]) continue;
            if (!bank_stream_supported(i + 1, F, E, L[i], n_atoms, x_stride, e_unit4[i])) streamed = false;
            const int nct = (L[i] + 15) / 16;
            coefq4[i] = (float*)(ws + coef_off);
            coef_off += (size_t)((buckets[i].count + 15) / 16) * nct * 512 * 4;
        }
    }
    if (through_nei) {
        // grad_out is the gradient of h = propagate(out): only the streamed pair folds that step in (its pre-pass sums the
        // neighbours' rows); every degree with atoms and kernels must be on it
        bool all = streamed && rows_streamed;
        for (int i = 0; i < 4; ++i) if (buckets[i].count > 0 && L[i] > 0 && !bank_use[i]) all = false;
        if (!all) return fail("%s: MKGNN_BACKWARD_THROUGH_NEIGHBOURS needs the streamed kernels for every degree "
                              "(mkgnn_backward_streams(..) tells)", who);
    }
    if (rows_split) {
        // pre-split rows: only the streamed bank kernel and the pipelined gather read them
        bool all = streamed && rows_streamed && !force_generic && bank_stream_rows_split_supported(F) &&
                   (!grad_x || (grad_x_stride % 4 == 0 && ((uintptr_t)grad_x & 15) == 0));
        for (int i = 0; i < 4; ++i) if (buckets[i].count > 0 && L[i] > 0 && !bank_use[i]) all = false;
        if (!all) return fail("%s: MKGNN_BACKWARD_ROWS_SPLIT needs the streamed kernels with split-fp16 products for every degree "
                              "(mkgnn_rows_split_supported(..) tells)", who);
    }
    BankStreamLaunch bsl;
    int nchunk4[4] = {0, 0, 0, 0}, ntheta4[4] = {0, 0, 0, 0};
    if (streamed) {
        // the pre-pass first, on the caller's stream: its records (dL/dsc and permutation ids in tile order) feed both
        // the rows kernel here and the bank kernel on the helper
        plan_backward_bank_stream(bank_a, bank_use, e_unit4, coefq4, nchunk4, ntheta4, through_nei, &bsl);
        bsl.x_split = rows_split ? 1 : 0;
        { BwdTimer t(st, 0); e = launch_coef_prepare(bsl, st); }
        if (e != hipSuccess) return hip_fail("coefficient pre-pass launch", e);
        e = fj.refork(1);
        if (e != hipSuccess) return hip_fail("stream fork", e);
    }
    if (rows_streamed && any_bank && grad_x) {
        BwdTimer t(st, 1);
        e = launch_backward_rows_stream(bank_a, bank_use, streamed ? coefq4 : nullptr, st);       // (bank_a holds every active degree's arguments)
        if (e != hipSuccess) return hip_fail("streamed rows launch", e);
    }
    if (any_bank) {
        hipStream_t st_bank = fj.stream(1, &e);      // the helper (the caller's stream when nothing is forked)
        if (e != hipSuccess) return hip_fail("stream fork", e);
        if (st_bank != st) {                         // a fused tail's deferred reduction rides in front of the bank kernel
            e = launch_pending_tail_reduce(st_bank);
            if (e != hipSuccess) return hip_fail("deferred tail reduction launch", e);
        }
        {
            BwdTimer t(st_bank, 2);
            if (streamed) e = launch_backward_bank_stream(bsl, st_bank);
            else e = launch_backward_bank_fused(bank_a, bank_use, nchunk4, ntheta4, st_bank);
        }
        if (e != hipSuccess) return hip_fail("fused bank gradient launch", e);
        for (int i = 0; i < 4; ++i)
            if (bank_use[i]) {
                reduce[i].nchunk = nchunk4[i];
                reduce[i].theta_src = bank_a[i].theta_slab; reduce[i].theta_stride = 4; reduce[i].theta_count = ntheta4[i];
            }
    }
    // two chains: the reduce follows the bank kernels on the helper, the gather follows the rows kernels here
    const bool split = fj.two_way && !bank_on_main && fj.used[0];
    if (!split) {
        e = fj.end();
        if (e != hipSuccess) return hip_fail("stream join", e);
    }
    hipStream_t st_reduce = st;
    if (split) {
        st_reduce = fj.stream(1, &e);
        if (e != hipSuccess) return hip_fail("stream fork", e);
    }
    { BwdTimer t(st_reduce, 3); e = launch_bank_reduce_all(reduce, st_reduce); }   // one launch for the four banks
    if (e != hipSuccess) return hip_fail("bank gradient reduce launch", e);
    if (grad_x) {
        BwdTimer t(st, 4);
        e = launch_backward_gather((const float*)(ws + w.contrib), (F + 3) / 4 * 4, base, scatter_rowptr, scatter_rows, x,
                                   x_stride, inv_norm, n_atoms, F, grad_x, grad_x_stride, !force_generic, st, rows_split);
        if (e != hipSuccess) return hip_fail("backward gather launch", e);
    }
    if (split && defer_bank && fj.used[0] && !fj.used[1] && !fj.used[2]) {
        fj.p->deferred = true;                       // the caller joins (mkgnn_backward_join), after the last layer's backward
    } else if (split) {
        e = fj.end();
        if (e != hipSuccess) return hip_fail("stream join", e);
    }
    return 0;
}

int mkgnn_backward_streams(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE], const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                           const float* x, int64_t x_stride, int64_t n_atoms, int32_t F, int32_t E) {
    if (!banks || !buckets) return 0;
    return streamed_pair_covers(banks, buckets, x, x_stride, n_atoms, F, E) ? 1 : 0;
}

int mkgnn_rows_split_supported(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE], const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                               int64_t x_stride, int64_t out_stride, int64_t n_atoms, int32_t F, int32_t E) {
    if (!banks || !buckets || F < 1 || E < 1) return 0;
    return rows_split_covered(banks, buckets, nullptr, x_stride, out_stride, n_atoms, F, E) ? 1 : 0;
}

int mkgnn_backward_join(void* stream) {
    DegreeStreams* p = degree_streams();
    if (!p || !p->deferred) return 0;
    p->deferred = false;
    hipError_t e = hipEventRecord(p->join[0], p->aux[0]);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, p->join[0], 0);
    return e == hipSuccess ? 0 : hip_fail("mkgnn_backward_join", e);
}

int mkgnn_segment_sum_rows(const float* in, int64_t in_stride, const int32_t* rowptr, const int32_t* col, int64_t n_rows,
                           int32_t width, float* out, int64_t out_stride, float* inv_norm, void* stream) {
    if (n_rows < 0 || width <= 0 || width > 16384 || in_stride < width || out_stride < width)
        return fail("mkgnn_segment_sum_rows: bad shape");
    if (n_rows && (!in || !rowptr || !out)) return fail("mkgnn_segment_sum_rows: null pointer");
    hipError_t e = launch_segment_sum(in, in_stride, rowptr, col, n_rows, width, out, out_stride, inv_norm, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail("mkgnn_segment_sum_rows", e);
}

int mkgnn_segment_sum_block_rows(const float* in, int64_t in_stride, const int32_t* rowptr, const int32_t* col,
                                 const int8_t* degree, int64_t n_rows, const int32_t num_kernels[MKGNN_MAX_DEGREE],
                                 int32_t mode, float* out, int64_t out_stride, float* inv_norm, void* stream) {
    const char* who = "mkgnn_segment_sum_block_rows";
    if (mode < 1 || mode > 3) return fail("%s: mode %d (1 = block-row sources, 2 = block-row destinations, 3 = 1 with pre-split output)", who, mode);
    if (mode == 3 && !inv_norm) return fail("%s: mode 3 (pre-split rows) needs inv_norm", who);
    if (!num_kernels) return fail("%s: num_kernels is null", who);
    int width = 0;
    for (int i = 0; i < MKGNN_MAX_DEGREE; ++i) {
        if (num_kernels[i] < 0 || num_kernels[i] > 255) return fail("%s: num_kernels[%d] = %d outside 0..255", who, i, num_kernels[i]);
        width += num_kernels[i];
    }
    if (n_rows < 0 || width <= 0 || in_stride < width || out_stride < width) return fail("%s: bad shape", who);
    if (n_rows == 0) return 0;
    if (!in || !rowptr || !col || !out || (mode == 2 && !degree)) return fail("%s: null pointer", who);
    if (!segment_sum_blocks_supported(in, in_stride, n_rows, width, out, out_stride))
        return fail("%s: needs 16-byte aligned rows (strides multiples of 4 floats), at most 255 columns and fewer than 2^28 rows", who);
    hipError_t e = launch_segment_sum_blocks(in, in_stride, rowptr, col, degree, n_rows, width, num_kernels, mode, out, out_stride,
                                             inv_norm, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(who, e);
}

}  // extern "C"
