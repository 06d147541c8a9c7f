// The parameter-only preparation of the kernel banks as device code two launches share (round 6): mkgnn_bank_prepare's own kernel
// (kgnn_generic.hip) and -- a preparation left pending by mkgnn_bank_prepare_deferred -- blocks behind the batch norm's statistics
// launch (kgnn_readout.hip), where the 5 us this launch costs the step's critical chain are hidden.
#pragma once
#include "kgnn_common.h"
#include "kgnn_launch.h"

namespace mkgnn {

// Unit-normalise every kernel row, keep 1/norm, tabulate the support tetrahedron signs and the mixing weights.  One wave per task.
__device__ __forceinline__ void bank_prepare_task(const PrepArgs& a, const int task) {
    const int lane = threadIdx.x & 63;
    int i = 0;
    while (task >= a.row_start[i + 1]) ++i;
    const int d = i + 1;
    const int L = a.bank[i].num_kernels;
    int r = task - a.row_start[i];
    // task order inside a degree: L centre rows, L*d support rows, L*d edge rows, 1 misc task, degree 4: ceil(12 L / 64) table tasks
    const float* src;
    float* dst;
    float* inv;
    int width;
    float* pad_dst = nullptr;
    int pad_width = 0;
    const int FP = bank_pitch(a.F);
    if (r < L) {
        src = a.bank[i].x_center + (size_t)r * a.F; dst = a.cen[i] + (size_t)r * a.F; inv = a.icen[i] + r; width = a.F;
        if (FP) { pad_dst = a.padded[i] + ((size_t)d * L + r) * FP; pad_width = FP; }
    } else if (r < L + L * d) {
        r -= L;
        src = a.bank[i].x_support + (size_t)r * a.F; dst = a.sup[i] + (size_t)r * a.F; inv = a.isup[i] + r; width = a.F;
        if (FP) { pad_dst = a.padded[i] + ((size_t)(r % d) * L + r / d) * FP; pad_width = FP; }
    } else if (r < L + 2 * L * d) {
        r -= L + L * d;
        src = a.bank[i].edge_attr_support + (size_t)r * a.E; dst = a.edg[i] + (size_t)r * a.E; inv = a.iedg[i] + r; width = a.E;
        if (a.E <= 8) { pad_dst = a.edge_padded[i] + ((size_t)(r % d) * L + r / d) * 8; pad_width = 8; }
    } else if (r > L + 2 * L * d) {
        // chirality table (kernels.py:331-341), 64 entries per task: every entry is a chain of dependent loads
        const int t = (r - (L + 2 * L * d) - 1) * 64 + lane;
        if (d == 4 && a.bank[i].p_support != nullptr && t < L * 12) {
            const int l = t / 12, p = t % 12;
            const float* ps = a.bank[i].p_support + (size_t)l * 12;   // [4, 3]
            a.chir[i][t] = (int8_t)triple_sign(ps + 3 * PERM4[p][0], ps + 3 * PERM4[p][1], ps + 3 * PERM4[p][2]);
        }
        return;
    } else {
        // misc: mixing weights (kernels.py:402-412)
        if (lane == 0) {
            float es = expf(*a.bank[i].support_attr_sc_weight);
            float ec = expf(*a.bank[i].center_attr_sc_weight);
            float ee = expf(*a.bank[i].edge_attr_support_sc_weight);
            float den = __fadd_rn(__fadd_rn(es, ec), ee);
            float ws = es / den, wc = ec / den, we = ee / den;
            a.mix[i][0] = ws; a.mix[i][1] = wc; a.mix[i][2] = we;
            a.mix[i][3] = __fadd_rn(__fadd_rn(ws, wc), we);
        }
        return;
    }
    if (width <= 128) {
        // the model's rows (F = 28 / 110, E = 7): both halves of the row loaded at once, every output from registers
        // (the general loop below reads the row three times, one dependent round trip each)
        const float v0 = src[lane < width ? lane : 0], v1 = src[lane + 64 < width ? lane + 64 : 0];
        const float m0 = lane < width ? v0 : 0.f, m1 = lane + 64 < width ? v1 : 0.f;
        float s = fmaf(m0, m0, 0.f);                 // (same order as the loop: element lane, then lane + 64)
        if (width > 64) s = fmaf(m1, m1, s);
        s = wave_sum(s);
        const float iv = 1.f / fmaxf(sqrtf(s), MKGNN_EPS);
        if (lane < width) dst[lane] = m0 * iv;
        if (lane + 64 < width) dst[lane + 64] = m1 * iv;
        if (lane == 0) *inv = iv;
        if (pad_dst) {
            if (lane < pad_width) pad_dst[lane] = m0 * iv;
            if (lane + 64 < pad_width) pad_dst[lane + 64] = m1 * iv;
        }
        return;
    }
    float s = 0.f;
    for (int f = lane; f < width; f += 64) {
        float v = src[f];
        s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    float iv = 1.f / fmaxf(sqrtf(s), MKGNN_EPS);
    for (int f = lane; f < width; f += 64) dst[f] = src[f] * iv;
    if (lane == 0) *inv = iv;
    // padded, support-major copies for the MFMA kernels
    if (pad_dst) {
        for (int f = lane; f < pad_width; f += 64) pad_dst[f] = f < width ? src[f] * iv : 0.f;
    }
}


// block `blk` (256 threads = four tasks) of the preparation of m.count forward calls
__device__ __forceinline__ void bank_prepare_many_block(const PrepManyArgs& m, const int blk) {
    const int task = (blk * 256 + (int)threadIdx.x) >> 6;
    if (task >= m.task_start[m.count]) return;
    int k = 0;
    while (task >= m.task_start[k + 1]) ++k;
    bank_prepare_task(m.layer[k], task - m.task_start[k]);
}

}  // namespace mkgnn
