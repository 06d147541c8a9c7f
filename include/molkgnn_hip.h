/*
 * molkgnn_hip.h -- C ABI of the MI355X (gfx950) molecular-kernel convolution.
 *
 * This is the drop-in boundary for ONE path of LanceKnight/MolKGNN: the
 * degree-bucketed atom-neighbourhood x learnable-kernel similarity
 * (reference models/MolKGNN/kernels.py, KernelConv / KernelSetConv) and its
 * immediate caller's neighbour sum (models/MolKGNN/KernelLayer.py, MolGCN).
 * The reference has no FFI of its own (it is pure Python over ATen); the entry
 * points below are what a binding for this path would bind, one per reference
 * function, and INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless it says host;
 *   - floats are fp32, indices int64 exactly as the reference's tensors hold them;
 *   - the caller owns every buffer (inputs, outputs, workspace); nothing is
 *     allocated or freed inside, no call synchronises, every kernel is launched
 *     on `stream` (a hipStream_t passed as void*), so calls are graph-capturable;
 *   - return 0 on success, non-zero on a rejected argument or a launch error;
 *     mkgnn_last_error() then describes it (thread-local, host pointer);
 *   - no state is kept between calls except lazily created helper streams / events (one set per device,
 *     created under std::call_once); calls on DIFFERENT devices or different streams of one device may run from
 *     different host threads, but mkgnn_kernelsetconv_backward forks onto the device's helper streams, so at
 *     most one host thread per device may be inside it at a time (PyTorch's autograd engine guarantees that:
 *     one backward thread per device).
 *
 * Shapes use the reference's names: N atoms in the batch, F node-attribute
 * width, E edge-attribute width, D = 3 coordinates, d = degree (1..4),
 * N_d atoms of degree d, L_d kernels of degree d, K = L_1+L_2+L_3+L_4.
 */
#ifndef MOLKGNN_HIP_H
#define MOLKGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MKGNN_MAX_DEGREE 4
#define MKGNN_ABI_VERSION 7

/* One KernelConv's parameters (reference kernels.py:50-84).  The three score
 * weights are the 0-d parameters support_attr_sc_weight, center_attr_sc_weight
 * and edge_attr_support_sc_weight; length_sc_weight / angle_sc_weight are never
 * read by the reference's forward (kernels.py:402-422) and are not passed. */
typedef struct mkgnn_kernel_bank {
    int32_t num_kernels;              /* L_d; 0 = this degree has no kernels       */
    int32_t reserved;
    const float* x_center;            /* [L_d, F]                                   */
    const float* x_support;           /* [L_d, d, F]                                */
    const float* edge_attr_support;   /* [L_d, d, E]                                */
    const float* p_support;           /* [L_d, d, 3]  (read only for d = 4, last layer) */
    const float* support_attr_sc_weight;      /* [1] */
    const float* center_attr_sc_weight;       /* [1] */
    const float* edge_attr_support_sc_weight; /* [1] */
} mkgnn_kernel_bank;

/* Gradients of one bank; same shapes.  p_support has no gradient in the
 * reference (kernels.py:279-350 is not differentiable) and has no slot. */
typedef struct mkgnn_kernel_bank_grad {
    float* x_center;
    float* x_support;
    float* edge_attr_support;
    float* support_attr_sc_weight;
    float* center_attr_sc_weight;
    float* edge_attr_support_sc_weight;
} mkgnn_kernel_bank_grad;

/* One degree's receptive fields, the tensors ToXAndPAndEdgeAttrForDeg produces
 * (reference wrapper.py:596-635) after PyG collation. */
typedef struct mkgnn_degree_bucket {
    int64_t count;                    /* N_d                                         */
    const int64_t* selected_index;    /* [N_d]      focal atom ids                   */
    const int64_t* nei_index;         /* [N_d * d]  neighbour atom ids, edge-list order */
    const float* nei_edge_attr;       /* [N_d, d, E] raw bond attributes             */
    const float* p_focal;             /* [N_d, 3]   (d = 4, last layer only; else may be NULL) */
    const float* nei_p;               /* [N_d, d, 3] (same)                          */
    const float* nei_edge_unit;       /* [N_d, d, 8] nei_edge_attr / max(|.|, 1e-8), zero padded (mkgnn_unit_rows8; E <= 8);
                                         may be NULL.  The bond attributes of a batch are the same in every layer and every
                                         step: a caller that keeps them (molkgnn_amd.plan does) saves the forward kernel the
                                         per-tile normalisation, and lets it take the streamed kernel */
} mkgnn_degree_bucket;

/* What forward keeps for backward, per degree (caller-allocated). */
typedef struct mkgnn_saved {
    float* pair_state;                /* [N_d, L_d, 4] per (atom, kernel) pair, one 16-byte record: the support score of
                                         the chosen permutation, the centre score, the edge score (kernels.py:357-373,
                                         386-395) and the chosen permutation's index (the bits of an int32).  One record
                                         is one 16-byte store in the forward and one 16-byte load in the backward;
                                         may be NULL (nothing is kept) */
    int8_t* chirality;                /* [N_d, L_d] +1/-1 (d = 4, last layer); may be NULL           */
} mkgnn_saved;

/* ABI history.  v3: pair records (mkgnn_saved.pair_state).  v4: the AdamW state buffer of a tensor is
 * mkgnn_adamw_state_floats(numel) = 2 numel + 3 + ceil(numel / 1024) floats (v3: 2 numel + 3) -- a caller built against
 * v3 would hand mkgnn_adamw_step a buffer its blocks write past, so the version check must refuse it; the
 * molecule-resident small-batch entry points (mkgnn_molecule_*) were added with the same version.  v5: the statistics-only
 * batch-norm companion (mkgnn_bn_stats, mkgnn_batchnorm_update_stats, mkgnn_batchnorm_forward_with_stats).  v6: pre-split rows
 * (MKGNN_VARIANT_ROWS_SPLIT / MKGNN_BACKWARD_ROWS_SPLIT / MKGNN_BN_SPLIT_ROWS, mkgnn_rows_presplit, mode 3 of
 * mkgnn_segment_sum_block_rows) and the fused tail (mkgnn_tail_*).  v7: mkgnn_touch_hint. */
int mkgnn_abi_version(void);
const char* mkgnn_last_error(void);

/* 1 / max(||x_n||, 1e-8) for every row (the cosine normalisation of
 * torch.nn.CosineSimilarity as used at kernels.py:189).  inv_norm: [N]. */
int mkgnn_row_inv_norm(const float* x, int64_t x_stride, int64_t n_rows, int32_t F,
                       float* inv_norm, void* stream);

/* out[r, 0..7] = in[r, 0..E-1] / max(||in[r]||, 1e-8), zero beyond E (E <= 8): the unit bond-attribute rows
 * mkgnn_degree_bucket.nei_edge_unit holds.  in: [n_rows, E] contiguous, out: [n_rows, 8]. */
int mkgnn_unit_rows8(const float* in, int64_t n_rows, int32_t E, float* out, void* stream);

/* Bytes of scratch the two calls below need for these sizes. */
size_t mkgnn_workspace_bytes(const int32_t num_kernels[MKGNN_MAX_DEGREE], int32_t F, int32_t E,
                             int64_t n_atoms, int64_t n_edges);

/* BaseKernelSetConv.forward (kernels.py:610-751) with the per-degree gather
 * (:519-548), KernelConv.calculate_total_score (:353-425) and the reorder to
 * node order (:743-747) fused: out[n, :] for atom n holds the L_d scores of its
 * own degree in columns [off_d, off_d + L_d) and zeros elsewhere.
 *   x        [N, F] with row stride x_stride (floats)
 *   inv_norm [N] from mkgnn_row_inv_norm
 *   out      [N, K] with row stride out_stride, fully overwritten (rows of atoms in no bucket
 *            are zero).  If out_stride holds K rounded up to a multiple of 4, that alignment
 *            padding is zeroed too, so the next layer can read 16-byte rows.
 * variant: 0 = automatic, 1 = generic VALU kernels, 2 = MFMA kernels (exact fp32),
 *          3 = MFMA kernels with bf16 operands for the node-feature dot products (fp32 accumulate; the
 *              "bf16 similarity path": scores within ~1e-3 of the fp32 ones, the chosen order may differ
 *              between near-tied permutations).  2 and 3 fail if a degree's shape is not covered.
 *          | MKGNN_VARIANT_BLOCK_ROWS: the caller will read only each atom's own column block (as
 *              mkgnn_segment_sum_block_rows does): nothing outside the blocks is written, not even zeros. */
#define MKGNN_VARIANT_BLOCK_ROWS 0x100
/*          | MKGNN_VARIANT_BANK_PREPARED: `workspace` already holds this call's normalised kernel bank, written by
 *              mkgnn_bank_prepare for these banks, F and E since the parameters last changed. */
#define MKGNN_VARIANT_BANK_PREPARED 0x200
/*          | MKGNN_VARIANT_ROWS_SPLIT (ABI v6): the rows of `x` are PRE-SPLIT -- written by mkgnn_segment_sum_block_rows mode 3 (or
 *              mkgnn_batchnorm_forward* with MKGNN_BN_SPLIT_ROWS): every four consecutive floats x[4 g .. 4 g + 3] of a row are stored
 *              as the sixteen bytes  fp16 hi(0..3) | fp16 lo(0..3)  of x * 2^(exponent(inv_norm[n]) + 8), hi = fp16(.), lo =
 *              fp16(. - hi) -- the operand the streamed kernels' matrix instructions take, so that no wave converts a row again
 *              (same bytes per row, same strides; the forward's scores are bit for bit those of the fp32 rows).  Such rows
 *              exist only between two calls of this library (the reference's h = propagate(sim_sc), KernelLayer.py:119-123,
 *              kernels.py:527,543).  Needs mkgnn_rows_split_supported(..) == 1; fails otherwise. */
#define MKGNN_VARIANT_ROWS_SPLIT 0x400
/* Rows from OUTSIDE the library in that form (benchmarks, tests; a training step's rows come pre-split from their producers):
 * inv_norm[n] = 1 / max(|x[n]|, 1e-8) exactly as mkgnn_row_inv_norm, out[n] = the pre-split row.  16-byte aligned rows, F <= 256. */
int mkgnn_rows_presplit(const float* x, int64_t x_stride, int64_t n_rows, int32_t F, float* inv_norm, float* out,
                        int64_t out_stride, void* stream);
/* The parameter-only part of `count` (<= 4 per call) forward calls -- unit-normalised kernel rows in the layouts the
 * kernels read, their norms, the chirality sign tables, the mixing weights: reference kernels.py:189, 279-350,
 * 386-395 -- in ONE launch: call k has banks[4 k .. 4 k + 3], feature width F[k], and gets the head of workspaces[k]
 * (a buffer of at least mkgnn_workspace_bytes(..) for that call) filled.  The banks depend on the parameters only, so
 * a model prepares all its layers at the start of a step instead of one small dependent launch per layer. */
int mkgnn_bank_prepare(int32_t count, const mkgnn_kernel_bank* banks, const int32_t* F, int32_t E,
                       void* const* workspaces, const size_t* workspace_bytes, void* stream);
/* The same preparation left PENDING on the current device (ABI v7): nothing is launched; the next mkgnn_batchnorm_forward*
 * (training mode) on this device carries its tasks in blocks behind its own statistics launch -- a launch bound by latency, where
 * they cost nothing, instead of 5 us of launch floor on the chain in front of the first convolution -- or
 * mkgnn_bank_prepare_flush(stream) launches it on its own, whichever comes first (mkgnn_bank_prepare flushes a pending one
 * too).  banks, F and the workspaces are read at the LAUNCH: they must stay as they are until then.  The caller MUST flush (or
 * mkgnn_bank_prepare_withdraw, which returns 1 if one was pending) before the first mkgnn_kernelsetconv_forward with
 * MKGNN_VARIANT_BANK_PREPARED, and before the workspaces go.  One pending preparation per device. */
int mkgnn_bank_prepare_deferred(int32_t count, const mkgnn_kernel_bank* banks, const int32_t* F, int32_t E,
                                void* const* workspaces, const size_t* workspace_bytes);
int mkgnn_bank_prepare_flush(void* stream);
int mkgnn_bank_prepare_withdraw(void);
/* A hint (ABI v7): `count` (<= 16) device arrays that the NEXT call of this host thread to mkgnn_bank_prepare or to
 * mkgnn_batchnorm_forward* (training mode) READS once and discards, with spare blocks of a launch it makes anyway -- the index
 * arrays of the batch the convolutions that follow will gather through (mkgnn_degree_bucket.selected_index / nei_index /
 * nei_edge_unit, the CSR of propagate).  Nothing is computed from them and no result depends on it: after a backward pass has
 * moved a gigabyte through the caches they are cold, and the first convolution of a step -- the one with the least matrix work
 * per fetched row to hide a miss behind -- pays 8-9 us of its 30 for them at the benchmark batch; read beside the batch norm's
 * statistics (a launch bound by latency, not by bandwidth) they cost nothing.  The arrays must stay allocated until that next
 * call's launch has run.  mkgnn_touch_hint(NULL, NULL, 0) withdraws a hint nobody took; returns 1 if there was one, else 0. */
int mkgnn_touch_hint(const void* const* arrays, const size_t* bytes, int32_t count);

int mkgnn_kernelsetconv_forward(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE],
                                const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                                const float* x, int64_t x_stride, const float* inv_norm,
                                int64_t n_atoms, int32_t F, int32_t E, int32_t is_last_layer,
                                float* out, int64_t out_stride,
                                const mkgnn_saved saved[MKGNN_MAX_DEGREE],
                                void* workspace, size_t workspace_bytes, int32_t variant, void* stream);

/* Gradient of the call above (what autograd derives for the reference).
 *   grad_out  [N, K] (stride grad_out_stride)
 *   scatter   CSR over atoms of the contribution rows that belong to each atom:
 *             contribution rows are numbered bucket by bucket, atom by atom,
 *             (focal, neighbour 0, .., neighbour d-1); scatter_rowptr [N+1],
 *             scatter_rows [sum_d N_d (d+1)] (int32).  Built once per batch by
 *             the host (molkgnn_amd.plan).
 *   grad_x    [N, F] (stride grad_x_stride), fully overwritten
 *   grads     per-degree parameter gradients, fully overwritten.
 *   workspace_from_forward  non-zero: `workspace` is the buffer the forward call of this layer used,
 *             untouched since, with the same banks / shapes: the normalised kernel bank in it is reused
 *             instead of being recomputed.
 *   variant   0 = automatic (the MFMA rows / LDS bank / pipelined gather kernels wherever a degree's shape is
 *             covered, the generic kernels elsewhere), 1 = the generic one-wave-per-atom kernels and the plain
 *             gather for every degree (the A/B reference of the parity tests), 2 = the fast kernels, failing if a
 *             degree with atoms and kernels is not covered by them.
 *             | MKGNN_BACKWARD_DEFER_BANK: `grad_x` is complete in stream order as always, but the kernel-bank
 *             gradients (`grads`) are left running on the device's helper stream when the call returns: they -- and the
 *             buffers their kernels read: x, inv_norm, grad_out, saved, the buckets, the workspace -- must not be
 *             touched until mkgnn_backward_join(stream) has been called.  The next layer's backward (which needs only
 *             grad_x) then overlaps this layer's bank gradients instead of waiting for them: nothing but the optimiser
 *             reads a weight gradient.  Honoured where the call forks at all (inside a hipGraph capture with the fast
 *             kernels); elsewhere the call is complete on return as without the flag and the join is a no-op. */
#define MKGNN_BACKWARD_DEFER_BANK 0x100
/*             | MKGNN_BACKWARD_THROUGH_NEIGHBOURS: `grad_out` is not the gradient of this call's `out` but of
 *             h = propagate(out) (KernelLayer.py:119-123: h[i] = sum over edges j -> i of out[j]), [N, K] like it.  The
 *             gradient of out[n, l] is then the sum of grad_out[m, l] over the targets m of n's edges, which are exactly
 *             the bucket's nei_index entries of n: the kernels' pre-pass sums those d rows itself and the caller skips
 *             the propagate step's own gradient pass.  Needs the streamed kernels for every degree
 *             (mkgnn_backward_streams); fails otherwise. */
#define MKGNN_BACKWARD_THROUGH_NEIGHBOURS 0x200
/*             | MKGNN_BACKWARD_ROWS_SPLIT (ABI v6): `x` is pre-split as in the forward call (MKGNN_VARIANT_ROWS_SPLIT): the bank
 *             kernel takes the halves as they are, the gather that undoes the row normalisation reads x / |x| as (hi + lo)
 *             mantissa(inv_norm) 2^-8.  grad_x is ordinary fp32.  Needs mkgnn_rows_split_supported(..) == 1. */
#define MKGNN_BACKWARD_ROWS_SPLIT 0x400
int mkgnn_kernelsetconv_backward(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE],
                                 const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                                 const float* x, int64_t x_stride, const float* inv_norm,
                                 int64_t n_atoms, int32_t F, int32_t E, int32_t is_last_layer,
                                 const float* grad_out, int64_t grad_out_stride,
                                 const mkgnn_saved saved[MKGNN_MAX_DEGREE],
                                 const int32_t* scatter_rowptr, const int32_t* scatter_rows,
                                 float* grad_x, int64_t grad_x_stride,
                                 const mkgnn_kernel_bank_grad grads[MKGNN_MAX_DEGREE],
                                 void* workspace, size_t workspace_bytes, int32_t workspace_from_forward,
                                 int32_t variant, void* stream);

/* 1 when a backward call with these banks, buckets and x would run every degree that has atoms and kernels on the
 * streamed kernels (the condition of MKGNN_BACKWARD_THROUGH_NEIGHBOURS), else 0.  Launches nothing. */
int mkgnn_backward_streams(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE],
                           const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                           const float* x, int64_t x_stride, int64_t n_atoms, int32_t F, int32_t E);

/* 1 when both the forward and the backward call with these banks, buckets and strides take pre-split rows
 * (MKGNN_VARIANT_ROWS_SPLIT / MKGNN_BACKWARD_ROWS_SPLIT): every degree with atoms and kernels on the streamed kernels with the
 * split-fp16 products (rows of at most 112 floats, 16-byte aligned, unit bond rows present).  Launches nothing. */
int mkgnn_rows_split_supported(const mkgnn_kernel_bank banks[MKGNN_MAX_DEGREE],
                               const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE],
                               int64_t x_stride, int64_t out_stride, int64_t n_atoms, int32_t F, int32_t E);

/* Makes `stream` wait for every bank-gradient chain that calls with MKGNN_BACKWARD_DEFER_BANK left on this device's
 * helper stream (no-op when there is none).  Graph-capturable: captured, it is the edge that joins the helper branch. */
int mkgnn_backward_join(void* stream);

/* MolGCN.propagate with aggr='add' (KernelLayer.py:14,119-123) as a CSR segment
 * sum: out[i, :] = sum_{k in [rowptr[i], rowptr[i+1])} in[col[k], :].
 * Forward uses the edges grouped by target (col = sources); the gradient is the
 * same call on the edges grouped by source (col = targets).  width <= strides.
 * 16-byte aligned rows (bases, strides % 4 == 0 and >= width rounded up to 4, width <= 256) take the
 * pipelined kernel, which also writes the alignment padding of `out` as zero.
 * inv_norm (may be NULL): also write 1 / max(||out[i]||, 1e-8) -- the next layer's cosine needs it,
 * so the producer of h hands it over instead of a separate pass over h. */
int mkgnn_segment_sum_rows(const float* in, int64_t in_stride, const int32_t* rowptr,
                           const int32_t* col, int64_t n_rows, int32_t width,
                           float* out, int64_t out_stride, float* inv_norm, void* stream);

/* The same sum where one side is the [N, K] output of a kernel convolution, whose row n is non-zero only in the
 * column block of atom n's degree (K = sum of num_kernels, block d = columns [off_d, off_d + num_kernels[d-1])):
 *   mode 1  the gathered rows `in` are block rows (MolGCN.propagate's forward: h = sum of neighbours' sim_sc).  col
 *           entries carry the source atom's degree (0 = in no bucket, contributes nothing) in bits 28..30; only the
 *           source's block is read -- ~K / L_d times fewer bytes per edge -- and memory outside the blocks is never
 *           looked at (the producer may leave it unwritten: MKGNN_VARIANT_BLOCK_ROWS).  out: dense [n_rows, K],
 *           inv_norm as above.  Bit-identical to mkgnn_segment_sum_rows on the zero-filled input.
 *   mode 2  the written rows `out` are block rows (its backward: d sim_sc = sum of the targets' dense d h rows, of
 *           which the convolution's backward reads only atom n's own block).  degree: [n_rows] int8, 0..4; only
 *           that block of out[n] is summed and written, the rest of the row is left untouched.  inv_norm unused.
 *   mode 3  (ABI v6) mode 1 with `out` written PRE-SPLIT (MKGNN_VARIANT_ROWS_SPLIT above; inv_norm required): for a caller whose
 *           only reader of out is the next mkgnn_kernelsetconv_forward / _backward.
 * Rows must be 16-byte aligned (strides multiples of 4 floats); K <= 255; n_rows < 2^28. */
int mkgnn_segment_sum_block_rows(const float* in, int64_t in_stride, const int32_t* rowptr, const int32_t* col,
                                 const int8_t* degree, int64_t n_rows, const int32_t num_kernels[MKGNN_MAX_DEGREE],
                                 int32_t mode, float* out, int64_t out_stride, float* inv_norm, void* stream);

/* ---- the consumers either side of the convolution stack -------------------
 *
 * Readout, reference MolKGNNNet.py:144-146:
 *     out[g] = sum_{n in molecule g} lin2( keep[n] * swish( lin1(h[n]) ) )
 * evaluated as  W2 (sum_n keep[n] * swish(W1 h[n] + b1)) + |g| b2  (lin2 and the add-pool commute).
 * Atoms of one molecule are contiguous (PyG batches): mol_ptr[g] .. mol_ptr[g+1].
 * Limits: F <= 128, H <= 64, G <= 64; h rows 16-byte aligned, h_stride a multiple of 4 and
 * >= F rounded up to 4. */
typedef struct mkgnn_readout_params {
    const float* lin1_weight;   /* [H, F] graph_embedding_lin1.weight */
    const float* lin1_bias;     /* [H] or NULL */
    const float* lin2_weight;   /* [G, H] graph_embedding_lin2.weight */
    const float* lin2_bias;     /* [G] or NULL */
    int32_t F, H, G;
} mkgnn_readout_params;

/* Row stride (floats) of the `pre` and `pooled` buffers below: H rounded up to 32 or 64. */
int32_t mkgnn_readout_hidden_stride(int32_t H);
size_t mkgnn_readout_workspace_bytes(int32_t F, int32_t H, int32_t G, int64_t n_atoms, int64_t n_mols);

/* keep_scale: [n_atoms, H] dropout multipliers (0 or 1/(1-p)), NULL = no dropout.
 * Writes pre [n_atoms, HS] (lin1 output, kept for backward), pooled [n_mols, HS], out [n_mols, G]. */
int mkgnn_readout_forward(const mkgnn_readout_params* params, const float* h, int64_t h_stride,
                          int64_t n_atoms, const int32_t* mol_ptr, int64_t n_mols,
                          const float* keep_scale, float* pre, float* pooled,
                          float* out, int64_t out_stride, void* stream);

/* atom_mol: [n_atoms] molecule id of every atom.  grad_h (may be NULL) is fully overwritten; the four
 * parameter gradients (each may be NULL) are fully overwritten, summed in a fixed order. */
int mkgnn_readout_backward(const mkgnn_readout_params* params, const float* h, int64_t h_stride,
                           int64_t n_atoms, const int32_t* mol_ptr, const int32_t* atom_mol, int64_t n_mols,
                           const float* keep_scale, const float* pre, const float* pooled,
                           const float* grad_out, int64_t grad_out_stride,
                           float* grad_h, int64_t grad_h_stride,
                           float* grad_lin1_weight, float* grad_lin1_bias,
                           float* grad_lin2_weight, float* grad_lin2_bias,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The same readout fed by the BLOCK ROWS of the last kernel convolution instead of by h = propagate(sim)
 * (KernelLayer.py:119-123 followed by MolKGNNNet.py:144-146): both steps are linear and row n of sim is non-zero only in
 * the column block of atom n's degree, so  z[n] = W1[:, block(n)] sim[n, block(n)]  is taken first and the H-wide rows are
 * propagated,  pre = propagate(z) + b1.  Same sums, re-associated; no dense [N, K] h and no gradient of it.
 *   sim [n_atoms, K] block rows (as MKGNN_VARIANT_BLOCK_ROWS leaves them), K = sum num_kernels = params->F, num_kernels <= 64
 *   buckets: only count and selected_index are read (the projection works on 16-atom tiles of one degree bucket, on the
 *   matrix cores); in_* / out_*: mkgnn_plan_build's CSRs of edge_index by target / by source
 *   z, pre (forward), dz (backward): [n_atoms, HS], HS = mkgnn_readout_hidden_stride(H); pooled, gate_sum [n_mols, HS].
 *   gate_sum NULL (inference): pre holds propagate(z), WITHOUT b1.  gate_sum non-NULL (a backward will follow): the forward
 *   leaves  gate = keep_scale * swish'(pre + b1)  IN PLACE of pre and its per-molecule sums in gate_sum; the backward takes
 *   that buffer as `gate` (d loss / d pre = dA[molecule] * gate is formed on the fly inside the propagate^T pass; db1 comes
 *   from gate_sum) and needs neither keep_scale nor pre.  grad_sim: [n_atoms, K] block rows (every atom's own block is
 *   written), may be NULL.
 * Limits: K <= 255, every block <= 64 kernels, H <= 64, G <= 64.  Workspace of the backward:
 * mkgnn_readout_blocks_workspace_bytes. */
size_t mkgnn_readout_blocks_workspace_bytes(int32_t K, int32_t H, int32_t G, int64_t n_mols);
int mkgnn_readout_blocks_supported(int32_t F, int32_t H, int32_t G, const int32_t num_kernels[MKGNN_MAX_DEGREE]);
int mkgnn_readout_blocks_forward(const mkgnn_readout_params* params, const float* sim, int64_t sim_stride,
                                 const int32_t num_kernels[MKGNN_MAX_DEGREE],
                                 const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], int64_t n_atoms,
                                 const int32_t* in_rowptr, const int32_t* in_col, const int32_t* mol_ptr, int64_t n_mols,
                                 const float* keep_scale, float* z, float* pre, float* pooled, float* gate_sum,
                                 float* out, int64_t out_stride, void* stream);
int mkgnn_readout_blocks_backward(const mkgnn_readout_params* params, const float* sim, int64_t sim_stride,
                                  const int32_t num_kernels[MKGNN_MAX_DEGREE],
                                  const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], int64_t n_atoms,
                                  const int32_t* out_rowptr, const int32_t* out_col, const int32_t* mol_ptr,
                                  const int32_t* atom_mol, int64_t n_mols, const float* gate, const float* gate_sum,
                                  const float* pooled, const float* grad_out, int64_t grad_out_stride,
                                  float* dz, float* grad_sim, int64_t grad_sim_stride,
                                  float* grad_lin1_weight, float* grad_lin1_bias, float* grad_lin2_weight,
                                  float* grad_lin2_bias, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the tail of a training step, fused (ABI v6) ----------------------------------------------------------------------
 * Everything behind the last kernel convolution, forward AND backward:
 *     h = propagate(sim)                                  KernelLayer.py:119-123
 *     emb_g = pool_g( lin2( swish( lin1(h) ) ) )          MolKGNNNet.py:144-146   (no dropout inside the readout)
 *     loss = mean_g BCEWithLogits( ffn( dropout(emb_g) ), target_g )     model.py:147-150, 169, 190-198; data.py:37
 * and, for d loss = 1, d loss / d sim (block rows: only every atom's own column block is written -- what the last
 * convolution's backward reads) and the six parameter gradients.  The loss is a mean over molecules, so a molecule's chain
 * back to d loss / d z is taken while it sits in LDS: FOUR launches -- z = W1 sim on the matrix cores; one kernel over chunks
 * of whole molecules for propagate, swish, pool, lin2, head, loss, their gradients and propagate^T; d sim = W1^T d z with the
 * dW1 partials on the matrix cores; one fixed-order reduction of all partial slabs -- in place of the nine of
 * mkgnn_readout_blocks_forward + mkgnn_bce_head_fused + mkgnn_readout_blocks_backward.  Same formulas, same re-associations
 * (project before propagate, pool before lin2), the same dropout mask as mkgnn_bce_head_fused for the same rng_state.
 *   sim         [n_atoms, K] block rows of the last mkgnn_kernelsetconv_forward (MKGNN_VARIANT_BLOCK_ROWS), K = readout.F,
 *               16-byte aligned rows; buckets [MKGNN_MAX_DEGREE]: count and selected_index are read (as the block-row readout)
 *   in_* / out_*  the edges grouped by target (columns = sources) / by source (columns = targets), int32, as for
 *               mkgnn_segment_sum_rows; mol_ptr [n_mols + 1], atom_mol [n_atoms]: atoms of a molecule contiguous, edges inside
 *   n_loss_mols the leading molecules that enter the loss (the rest -- padding molecules -- get zero gradients)
 *   rng_state   {seed, offset} int64, read and advanced by one when dropout_p > 0; rng_used receives what was read
 *   emb         [n_mols, G] or NULL;  pred [n_loss_mols];  loss [1];  grad_sim [n_atoms, K] block rows; grad_* may be NULL
 * Limits: mkgnn_tail_supported (the block-row readout's shapes with H <= 32 and G <= 32); and NO MOLECULE with more than
 * MKGNN_TAIL_MAX_ATOMS atoms or MKGNN_TAIL_MAX_EDGES edges (each way) -- the caller, who knows the molecule sizes, checks;
 * a molecule that breaks the promise is skipped and the loss comes back NaN.  workspace: mkgnn_tail_workspace_bytes. */
#define MKGNN_TAIL_MAX_ATOMS 128
#define MKGNN_TAIL_MAX_EDGES 512
typedef struct mkgnn_tail_args {
    const float* sim; int64_t sim_stride;
    int32_t num_kernels[MKGNN_MAX_DEGREE];
    const mkgnn_degree_bucket* buckets;      /* [MKGNN_MAX_DEGREE] */
    const int32_t* in_rowptr; const int32_t* in_col;
    const int32_t* out_rowptr; const int32_t* out_col;
    const int32_t* mol_ptr; const int32_t* atom_mol;
    int64_t n_atoms, n_mols, n_loss_mols;
    mkgnn_readout_params readout;
    const float* head_weight;   /* [G] ffn.weight */
    const float* head_bias;     /* [1] or NULL */
    const float* target;        /* [n_loss_mols] */
    float dropout_p;
    int64_t* rng_state; int64_t* rng_used;
    float* emb; int64_t emb_stride;
    float* pred; float* loss;
    float* grad_sim; int64_t grad_sim_stride;
    float *grad_lin1_weight, *grad_lin1_bias, *grad_lin2_weight, *grad_lin2_bias, *grad_head_weight, *grad_head_bias;
    /* (ABI v7) != 0: the last launch -- the fixed-order reduction that writes loss, the six parameter gradients and advances the
     * dropout generator; nothing before the optimiser reads any of them -- is NOT made by this call but by the calling thread's next
     * mkgnn_kernelsetconv_backward that forks a helper stream (on that stream, in front of its bank kernel: off the chain the next
     * layer's gradient waits for), or by mkgnn_tail_flush, whichever comes first.  grad_sim is complete when this call's launches
     * are.  The caller MUST call mkgnn_tail_flush(stream) before anything reads those outputs on `stream`. */
    int32_t defer_reduce;
} mkgnn_tail_args;
int mkgnn_tail_supported(int32_t K, int32_t H, int32_t G, const int32_t num_kernels[MKGNN_MAX_DEGREE]);
size_t mkgnn_tail_workspace_bytes(int32_t K, int32_t H, int32_t G, int64_t n_atoms, int64_t n_mols);
int mkgnn_tail_fused(const mkgnn_tail_args* args, void* workspace, size_t workspace_bytes, void* stream);
/* Launches a reduction that a mkgnn_tail_fused call with defer_reduce left pending on this thread, on `stream` (no-op: none). */
int mkgnn_tail_flush(void* stream);

/* BatchNorm1d over atom rows, reference MolKGNNNet.py:115 (torch.nn.BatchNorm1d semantics: biased
 * variance for the normalisation, unbiased for running_var, running <- running + momentum (batch - running)).
 * training != 0: batch statistics, running_* (may be NULL) updated in place, save_* written.
 * training == 0: running statistics; save_* (may be NULL) receive mean and 1/sqrt(var + eps).
 * inv_norm (may be NULL; needs C <= 32, C % 4 == 0 and 16-byte aligned rows of x and out): also
 * 1 / max(||out row||, 1e-8) per row, bit-identical to mkgnn_row_inv_norm
 * on out -- the first kernel convolution reads the normalised features next (MolKGNNNet.py:115-117).
 * num_batches_tracked (may be NULL): incremented when training != 0 (BatchNorm1d's counter).
 * n_valid_rows (device scalar, may be NULL): only rows [0, *n_valid_rows) enter the batch statistics; the rest of the
 * n_rows rows is padding (normalised like any row, excluded from every sum) -- a batch padded to a fixed shape so that
 * one captured graph serves every batch (molkgnn_amd.padding) keeps the statistics of its real atoms.
 * training | MKGNN_BN_SPLIT_ROWS (ABI v6; with inv_norm): `out` is written PRE-SPLIT (MKGNN_VARIANT_ROWS_SPLIT above) for a caller
 * whose only reader of it is the first mkgnn_kernelsetconv_forward / _backward. */
#define MKGNN_BN_SPLIT_ROWS 2
size_t mkgnn_batchnorm_workspace_bytes(int32_t C);
int mkgnn_batchnorm_forward(const float* x, int64_t x_stride, int64_t n_rows, int32_t C,
                            const float* weight, const float* bias,
                            float* running_mean, float* running_var, float momentum, float eps,
                            int32_t training, float* out, int64_t out_stride,
                            float* save_mean, float* save_invstd, float* inv_norm, int64_t* num_batches_tracked,
                            const int64_t* n_valid_rows, void* workspace, size_t workspace_bytes, void* stream);
/* Statistics-only companion of a batch norm (ABI v5).  The reference runs edge_batch_norm(data.edge_attr) in every
 * forward (MolKGNNNet.py:116); its OUTPUT never reaches the kernel convolution (kernels.py:610-751 ignores edge_attr,
 * SURVEY 8 a-1), but in training mode the call moves the module's running_mean / running_var / num_batches_tracked, and
 * those buffers are state-dict contents.  This is that side effect without the normalised rows: BatchNorm1d's update
 * (biased batch variance -> unbiased for running_var, running <- running + momentum (batch - running), counter + 1).
 *   row_key / key_limit (both or neither; device pointers): row r counts iff row_key[r] < *key_limit -- for a batch padded
 *   to a fixed shape (molkgnn_amd.padding) the bond rows of real atoms are those whose source id edge_index[0][r] is
 *   below n_valid_atoms.
 * mkgnn_batchnorm_update_stats runs it alone (one launch up to 64 K padded elements, three above);
 * mkgnn_batchnorm_forward_with_stats = mkgnn_batchnorm_forward with the companion's rows summed by extra blocks of the same
 * launches (training != 0 only; companion may be NULL).  Workspace: mkgnn_batchnorm_stats_workspace_bytes(companion C). */
typedef struct mkgnn_bn_stats {
    const float* x; int64_t x_stride; int64_t n_rows; int32_t C;
    float* running_mean; float* running_var;      /* [C], updated in place (either may be NULL) */
    float momentum;
    int64_t* num_batches_tracked;                 /* may be NULL */
    const int64_t* row_key; const int64_t* key_limit;
} mkgnn_bn_stats;
size_t mkgnn_batchnorm_stats_workspace_bytes(int32_t C);
int mkgnn_batchnorm_update_stats(const mkgnn_bn_stats* stats, void* workspace, size_t workspace_bytes, void* stream);
int mkgnn_batchnorm_forward_with_stats(const float* x, int64_t x_stride, int64_t n_rows, int32_t C,
                            const float* weight, const float* bias,
                            float* running_mean, float* running_var, float momentum, float eps,
                            int32_t training, float* out, int64_t out_stride,
                            float* save_mean, float* save_invstd, float* inv_norm, int64_t* num_batches_tracked,
                            const int64_t* n_valid_rows, void* workspace, size_t workspace_bytes,
                            const mkgnn_bn_stats* companion, void* companion_workspace, size_t companion_workspace_bytes,
                            void* stream);
/* grad_x (may be NULL), grad_weight, grad_bias (may be NULL) are fully overwritten. */
int mkgnn_batchnorm_backward(const float* grad_out, int64_t grad_out_stride, const float* x, int64_t x_stride,
                             int64_t n_rows, int32_t C, const float* weight,
                             const float* save_mean, const float* save_invstd, int32_t training,
                             float* grad_x, int64_t grad_x_stride, float* grad_weight, float* grad_bias,
                             const int64_t* n_valid_rows, void* workspace, size_t workspace_bytes, void* stream);

/* Receptive-field builder: the per-degree tensors of a collated batch, reference
 * wrapper.py:559-672 (ToXAndPAndEdgeAttrForDeg) + PyG collation.  edge_index is [2, M] int64 (row 0 sources, row 1
 * targets), every bond stored as two consecutive directed edges with identical attributes (wrapper.py:152-156).
 *   mkgnn_rf_count  computes the out-degrees and writes the four bucket sizes N_1..N_4 to counts (device, int64[4]);
 *   mkgnn_rf_fill   (same workspace, untouched in between) fills, for d = 1..4, out[d-1]:
 *       selected_index [N_d] ascending atom ids, nei_index [N_d*d] edge targets in edge-list order,
 *       nei_edge_attr [N_d*d, E] attributes of bond 2*(e/2), p_focal [N_d, 3], nei_p [N_d*d, 3];
 *       nei_edge_unit [N_d*d, 8] (if not NULL and E <= 8) the same rows unit-normalised, as mkgnn_unit_rows8 gives them;
 *       out[d-1].count = rows allocated (>= N_d).  Atoms of out-degree 0 or > 4 are in no bucket.
 * Deterministic: integer atomics only choose slots, the <= 4 edge ids of an atom are sorted afterwards. */
size_t mkgnn_rf_workspace_bytes(int64_t n_atoms);
int mkgnn_rf_count(const int64_t* edge_index, int64_t n_atoms, int64_t n_edges,
                   void* workspace, size_t workspace_bytes, int64_t* counts, void* stream);
int mkgnn_rf_fill(const int64_t* edge_index, const float* p, const float* edge_attr,
                  int64_t n_atoms, int64_t n_edges, int32_t E, const void* workspace,
                  const mkgnn_degree_bucket out[MKGNN_MAX_DEGREE], void* stream);

/* Per-batch index plan (the host side keeps it with the batch, molkgnn_amd.plan): for every atom, in ascending original
 * order inside the atom's segment,
 *   scatter  the contribution rows of mkgnn_kernelsetconv_backward that point at it (rows numbered bucket by bucket, atom
 *            by atom, focal then neighbours): scatter_rowptr [N+1], scatter_rows [sum_d N_d (d+1)];
 *   in       the edges that end in it, by source atom: in_rowptr [N+1], in_col [M] (MolGCN.propagate's forward);
 *            in_col_packed (may be NULL): the same with the source's degree bucket in bits 28..30
 *            (mkgnn_segment_sum_block_rows mode 1);
 *   out      the edges that start from it, by target atom: out_rowptr [N+1], out_col [M] (its gradient);
 *   deg8     [N] the degree bucket of every atom (0 = in none).
 * No sort over the batch and no host synchronisation: counts with integer atomics, a scan, slots taken with integer
 * atomics, every segment sorted by one thread -- the result does not depend on the order the atomics ran in.
 * buckets: only count, selected_index and nei_index are read.  N < 2^28, M < 2^31. */
size_t mkgnn_plan_workspace_bytes(int64_t n_atoms, int64_t n_edges, int64_t n_rows);

/* Receptive fields AND index plan in one pass (ABI v5): mkgnn_rf_count + mkgnn_rf_fill + mkgnn_plan_build for a caller that knows the
 * bucket capacities up front (a batch padded to a fixed shape: molkgnn_amd.padding) -- one memset and six kernels instead of
 * two and nine, no host round trip (capturable).  Inputs as mkgnn_rf_fill; out[d-1].count = rows allocated for degree d, which
 * must equal the batch's number of atoms of out-degree d: counts[0..3] (device, int64[6]) receive the real numbers, counts[4]
 * the number of atoms that did not fit (0 for a well-formed call), counts[5] = 0; rows beyond a real size are zero-filled.
 * PRECONDITION: every capacity EQUALS the real count (what a fixed-shape batch guarantees).  With a capacity above the real count
 * the scatter CSR covers the ranked atoms only -- scatter_rowptr[n_atoms] < the allocated rows, the tail of scatter_rows is left
 * unwritten -- which is NOT what mkgnn_plan_build does with zero-filled rows; the caller checks counts[0..3] against its
 * capacities (receptive_field.check_sizes) before trusting the plan.  Outputs as
 * mkgnn_rf_fill (the 20 tensors of wrapper.py:559-672 + the unit bond rows) and as mkgnn_plan_build (contribution rows
 * numbered with the allocated row counts), entry for entry what the separate calls give.  rf_ready_event (a hipEvent_t, may be
 * NULL) is recorded on the stream where the receptive fields are complete (the plan's two kernels follow): the first
 * convolution of a step needs only those. */
size_t mkgnn_index_workspace_bytes(int64_t n_atoms, int64_t n_edges, int64_t n_rows);
int mkgnn_index_build(const int64_t* edge_index, const float* p, const float* edge_attr, int64_t n_atoms, int64_t n_edges,
                      int32_t E, const mkgnn_degree_bucket out[MKGNN_MAX_DEGREE],
                      int32_t* scatter_rowptr, int32_t* scatter_rows, int32_t* in_rowptr, int32_t* in_col,
                      int32_t* in_col_packed, int32_t* out_rowptr, int32_t* out_col, int8_t* deg8, int64_t* counts,
                      void* workspace, size_t workspace_bytes, void* rf_ready_event, void* stream);

/* A collated batch from its compact wire form (what a loader sends over PCIe, molkgnn_amd/shards.py) to the tensors the
 * reference's batch object holds (PyG collation, wrapper.py:152-156):
 *   bond_ij   [n_bonds, 2] int32 batch-local atom ids      -> edge_index [2, 2 n_bonds] int64: bond k as the consecutive
 *                                                             directed edges (i, j), (j, i);
 *   bond_attr [n_bonds, E] uint8 (0..255-valued features)  -> edge_attr [2 n_bonds, E] fp32, the same row for both directions;
 *   mol_ptr   [n_molecules + 1] int32 first atom of every molecule -> batch [n_atoms] int64 and atom_molecule [n_atoms] int32.
 * Any output may be NULL (skipped).  One launch. */
int mkgnn_expand_batch(const int32_t* bond_ij, const uint8_t* bond_attr, int64_t n_bonds, int32_t E,
                       const int32_t* mol_ptr, int64_t n_molecules, int64_t n_atoms,
                       int64_t* edge_index, float* edge_attr, int64_t* batch, int32_t* atom_molecule, void* stream);
int mkgnn_plan_build(const mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE], int64_t n_atoms, const int64_t* edge_index,
                     int64_t n_edges, int32_t* scatter_rowptr, int32_t* scatter_rows, int32_t* in_rowptr, int32_t* in_col,
                     int32_t* in_col_packed, int32_t* out_rowptr, int32_t* out_col, int8_t* deg8, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Tail of the training step for a single task (reference model.py:147-148, 190-198 with data.py:37):
 *     pred = graph_embedding @ ffn.weight[0] + ffn.bias;   loss = mean(BCEWithLogits(pred, target))
 * forward writes pred [n_rows] and loss [1]; backward takes d loss (one float on the device) and fully overwrites
 * grad_emb [n_rows, H] (may be NULL), grad_weight [H], grad_bias [1] (may be NULL).  Two launches each (block
 * partials, then their sum in a fixed order).  workspace: mkgnn_bce_head_workspace_bytes bytes, not shared by calls
 * that may run concurrently. */
size_t mkgnn_bce_head_workspace_bytes(int64_t n_rows, int32_t H);
int mkgnn_bce_head_forward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H,
                           const float* weight, const float* bias, const float* target,
                           float* pred, float* loss, void* workspace, size_t workspace_bytes, void* stream);
int mkgnn_bce_head_backward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H,
                            const float* weight, const float* target, const float* pred, const float* grad_loss,
                            float* grad_emb, int64_t grad_emb_stride, float* grad_weight, float* grad_bias,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The same with dropout on the embedding ahead of the product (reference model.py:150,169: nn.Dropout(ffn_dropout_rate)
 * before ffn), generated in the kernels: element (row, h) is zeroed with probability dropout_p in [0, 1), kept values
 * scaled by 1 / (1 - dropout_p).  Counter-based generator (Philox4x32-10) keyed by rng_state = {seed, offset} on the
 * device: forward uses the pair, copies it to rng_used (2 x int64, the backward regenerates the mask from it) and
 * advances offset by one, so a replayed graph draws a fresh mask every time.  dropout_p = 0: the functions above. */
int mkgnn_bce_head_dropout_forward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H,
                                   const float* weight, const float* bias, const float* target, float dropout_p,
                                   int64_t* rng_state, int64_t* rng_used, float* pred, float* loss,
                                   void* workspace, size_t workspace_bytes, void* stream);
int mkgnn_bce_head_dropout_backward(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H,
                                    const float* weight, const float* target, const float* pred,
                                    const float* grad_loss, float dropout_p, const int64_t* rng_used,
                                    float* grad_emb, int64_t grad_emb_stride, float* grad_weight, float* grad_bias,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* Forward of mkgnn_bce_head_dropout_forward AND the gradients of mkgnn_bce_head_dropout_backward for d loss = 1 in one
 * pass (two launches instead of four): the loss ends the graph, so its own gradient is 1 in a training step, and
 * d loss / d pred of a row needs only that row's pred.  grad_emb [n_rows, H] (may be NULL), grad_weight [H], grad_bias [1]
 * (may be NULL) are fully overwritten with the gradients a backward call with *grad_loss == 1 would give, bit for bit.
 * A caller whose d loss is something else scales them.  Same workspace, same dropout generator protocol. */
int mkgnn_bce_head_fused(const float* emb, int64_t emb_stride, int64_t n_rows, int32_t H,
                         const float* weight, const float* bias, const float* target, float dropout_p,
                         int64_t* rng_state, int64_t* rng_used, float* pred, float* loss,
                         float* grad_emb, int64_t grad_emb_stride, float* grad_weight, float* grad_bias,
                         void* workspace, size_t workspace_bytes, void* stream);

/* AdamW step over all trainable tensors of the model in one launch (reference model.py:368-385: torch.optim.AdamW,
 * two parameter groups -- kernel banks without weight decay).  Per tensor: param / grad [numel] fp32 contiguous,
 * state [mkgnn_adamw_state_floats(numel)] = exp_avg, exp_avg_sq, step count (as a float, advanced by this call), two reserved
 * floats, then one copy of the step count per 1024 elements (every block of the update advances its own: one launch per
 * step, no counting pass; a caller that sets the step count sets every copy).  Per group: the
 * learning rate either by value (lr_device NULL) or read from a device float at run time (so that a captured graph
 * follows a scheduler), betas, eps, decoupled weight_decay, maximize, grad_scale.  The update is torch's fused AdamW formula
 * (bias corrections 1 - beta^step).  At most 4 groups; any number of tensors (80 per launch). */
/* (ABI v7) n_items small float tensors, each into its destination, in one launch (ceil(numel / 1024) blocks each): the
 * data-parallel step's gradients into the flat buffer one collective sums (dp.FlatGradAllReduce), where torch._foreach_copy_
 * takes two multi-tensor launches.  Sources and destinations must not overlap; graph-capturable (pointers are baked in). */
typedef struct mkgnn_copy_item { float* dst; const float* src; int64_t numel; } mkgnn_copy_item;
int mkgnn_flat_copy(const mkgnn_copy_item* items, int32_t n_items, void* stream);

typedef struct mkgnn_adamw_tensor {
    float* param;
    const float* grad;
    float* state;
    int64_t numel;
    int32_t group;
    int32_t reserved;
    const float* active;   /* NULL, or a device float: the tensor is skipped (no update, step count not advanced) when it
                              reads 0 -- "no rank had a gradient for it this step" in a captured data-parallel step */
} mkgnn_adamw_tensor;
typedef struct mkgnn_adamw_group {
    const float* lr_device;
    float lr, beta1, beta2, eps, weight_decay;
    int32_t maximize;
    float grad_scale;      /* every gradient is multiplied by this first (1 / world size after a summing all-reduce; else 1) */
} mkgnn_adamw_group;
int64_t mkgnn_adamw_state_floats(int64_t numel);
int mkgnn_adamw_step(const mkgnn_adamw_tensor* tensors, int32_t n_tensors, const mkgnn_adamw_group* groups,
                     int32_t n_groups, void* stream);

/* ---- Molecule-resident small-batch step (ABI v4).
 *
 * At the reference's own batch sizes (README.md:81 `--batch_size 16`, data.py:236 default 17; BASELINE configs[0] / [2]) a
 * step of one launch per operator is a chain of ~30 dependent launches of a few microseconds each.  Nothing on the path
 * crosses a molecule boundary after the batch-norm statistics (edge_index is block-diagonal), so here ONE launch runs, for
 * a chunk of whole molecules per workgroup (at most 64 atoms, its rows resident in LDS):
 *
 *   node_batch_norm -> num_layers x (KernelSetConv -> propagate) -> lin1, swish, add-pool, lin2   (MolKGNNNet.py:115-146,
 *                                                                                                  KernelLayer.py:109-120)
 *   [-> dropout -> ffn -> BCEWithLogitsLoss                                                       (model.py:150, 169, 190-198)]
 *   [-> the backward of all of it, back to the batch norm's weight and bias]
 *
 * followed by one reduction launch that sums the per-workgroup partial gradients in a fixed order (no float atomics) and
 * undoes the unit normalisation of the kernel rows, and preceded by one preparation launch (unit-normalised kernel rows,
 * mixing weights, chirality tables of every layer; partial batch-norm statistics).  Same arithmetic as the per-operator
 * entry points above -- cosines of unit rows, every neighbour order scored ((c0+c1)+c2)+c3 then / d, strict '>' scan in
 * table order, chirality sign -- so the parity criteria of the per-operator path apply unchanged; results are not
 * bit-identical to it (other summation orders inside a dot product).
 *
 * Shapes taken: 1..4 layers; first layer's input width <= 32, every layer's kernel count K <= 112 with at most 64
 * kernels per degree and (L_1 + 2 L_2 + 3 L_3) + (L_1 + L_2 + L_3) <= 256, 5 L_4 <= 256; E <= 8; H, G <= 64; every
 * molecule <= 64 atoms, atoms of a molecule contiguous, every atom's degree <= 4 and its bonds stored in both
 * directions (as the reference stores them, wrapper.py:152-156), unit bond rows present.  mkgnn_molecule_supported says
 * whether a model shape is taken; the batch conditions are the caller's (molkgnn_amd.molecule checks them once per batch). */
#define MKGNN_MOLECULE_MAX_LAYERS 4
#define MKGNN_MOLECULE_MAX_ATOMS 64      /* atoms per chunk (and so per molecule) */
#define MKGNN_MOLECULE_MAX_MOLS 16       /* molecules per chunk */

typedef struct mkgnn_molecule_layer {
    mkgnn_kernel_bank bank[MKGNN_MAX_DEGREE];        /* this layer's KernelSetConv (kernels.py:759-778) */
    mkgnn_kernel_bank_grad grad[MKGNN_MAX_DEGREE];   /* backward: where the bank gradients go (a NULL x_center: skipped) */
    mkgnn_saved saved[MKGNN_MAX_DEGREE];             /* pair records [N_d, L_d, 4] (+ chirality, last layer, d = 4):
                                                        written by the forward, needed by the backward */
    int32_t F;                                       /* input width of the layer (x_dim, then the previous layer's K) */
    int32_t reserved;
    float* sim_out;                                  /* optional [n_atoms, sim_stride]: this layer's sim_sc (tests) */
    int64_t sim_stride;
} mkgnn_molecule_layer;

typedef struct mkgnn_molecule_net {
    int32_t num_layers, E;
    mkgnn_molecule_layer layer[MKGNN_MOLECULE_MAX_LAYERS];
    /* node_batch_norm (MolKGNNNet.py:26, 115) */
    const float* bn_weight; const float* bn_bias;    /* [F0] or NULL */
    float* bn_running_mean; float* bn_running_var;   /* [F0]; updated when bn_training */
    int64_t* bn_num_batches_tracked;                 /* or NULL */
    float bn_eps, bn_momentum;
    int32_t bn_training, reserved;
    float* grad_bn_weight; float* grad_bn_bias;      /* backward outputs (NULL: skipped) */
    /* readout (MolKGNNNet.py:144-146): lin1 [H, K_last], lin2 [G, H] */
    mkgnn_readout_params readout;
    float* grad_lin1_weight; float* grad_lin1_bias; float* grad_lin2_weight; float* grad_lin2_bias;
    /* head (model.py:149-150, 169): ffn [1, G], dropout in front of it */
    const float* ffn_weight; const float* ffn_bias;
    float* grad_ffn_weight; float* grad_ffn_bias;
    float head_dropout;                              /* 0 = none */
    int32_t reserved2;
    int64_t* rng_state;                              /* {seed, offset} of the head's dropout (offset advanced by the call) */
    int64_t* rng_used;                               /* optional: the {seed, offset} this call's mask was drawn with */
    /* edge_batch_norm's side effect (MolKGNNNet.py:116; ABI v5): when not NULL
     * (the caller leaves it NULL on a backward-only call), the bond rows' statistics -- 1..8192 rows, C <= 8; anything else
     * is an error: use mkgnn_batchnorm_update_stats -- are taken by one extra block of the preparation launch and the
     * module's buffers updated as mkgnn_batchnorm_update_stats would */
    const mkgnn_bn_stats* edge_stats;
} mkgnn_molecule_net;

typedef struct mkgnn_molecule_batch {
    int64_t n_atoms, n_mols, n_chunks;
    int64_t max_chunk_atoms;                         /* atoms of the largest chunk (<= 32: the half-size kernel variant) */
    const int32_t* chunk_mol_ptr;                    /* [n_chunks + 1] first molecule of every chunk */
    const int64_t* mol_atom_ptr;                     /* [n_mols + 1] first atom of every molecule */
    const int8_t* atom_degree;                       /* [n_atoms] out-degree, 0..4 */
    const int32_t* atom_rank;                        /* [n_atoms] position of the atom in its degree bucket */
    mkgnn_degree_bucket buckets[MKGNN_MAX_DEGREE];   /* nei_index, nei_edge_unit; p_focal / nei_p for d = 4 */
    const float* x; int64_t x_stride;                /* [n_atoms, F0] raw node features */
} mkgnn_molecule_batch;

#define MKGNN_MOLECULE_HEAD      1   /* dropout -> ffn -> BCE-with-logits against `target`; writes pred, loss */
#define MKGNN_MOLECULE_BACKWARD  2   /* gradients of every parameter (d loss = 1 with HEAD, else from grad_emb) */
#define MKGNN_MOLECULE_GRAD_EMB  4   /* BACKWARD without HEAD: d loss / d graph embedding is given */

int mkgnn_molecule_supported(const mkgnn_molecule_net* net, int32_t x_dim);
size_t mkgnn_molecule_workspace_bytes(const mkgnn_molecule_net* net, int32_t x_dim, int64_t n_atoms, int64_t n_chunks);
/* emb [n_mols, G] is always written.  target [n_mols] and pred [n_mols] / loss [1] with HEAD; grad_emb [n_mols, G] with
 * GRAD_EMB.  The workspace is scratch between calls (the backward of a call recomputes what it needs). */
int mkgnn_molecule_step(const mkgnn_molecule_net* net, const mkgnn_molecule_batch* batch, int32_t mode,
                        const float* target, const float* grad_emb, float* emb, float* pred, float* loss,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- host side of the packed-shard loader (no device work): a fixed-shape batch of molecules [m0, m1) of a shard in the
 * compact wire form that mkgnn_expand_batch takes -- features and coordinates as they are, every bond once as an int32
 * pair with byte-valued attributes, labels, molecule pointers; padding atoms / bonds / molecules as molkgnn_amd.padding
 * defines them.  Replaces PyG's Python collation (reference data.py:168-203: DataLoader workers) on the loader threads;
 * one call per batch, so a Python caller's interpreter lock is released for all of it.  `shape` = {atoms, directed edges,
 * N_1, N_2, N_3, N_4} of the epoch's common shape; out: the staging buffer (mkgnn_collate_compact_bytes, fields 256-byte
 * aligned in the order x, p, bond_ij, bond_attr, y, mol_ptr, n_valid_atoms).  The shard's bonds must be stored as reversed
 * pairs (i, j), (j, i) with shared byte-valued attributes (the shard writer's `compact` flag). */
typedef struct mkgnn_shard_view {
    const float* x;                 /* [n_atoms, x_dim] */
    const float* p;                 /* [n_atoms, p_dim] */
    const int32_t* edge_src;        /* [n_edges] shard-global atom ids */
    const int32_t* edge_dst;
    const float* edge_attr;         /* [n_edges, e_dim] */
    const float* y;                 /* [n_molecules] */
    const int64_t* mol_atom_ptr;    /* [n_molecules + 1] */
    const int64_t* mol_edge_ptr;    /* [n_molecules + 1] */
    const int64_t* mol_deg_ptr;     /* [n_molecules + 1, 5] prefix sums of the atoms of degree 1..4 / in no bucket */
    int64_t n_molecules;
    int32_t x_dim, p_dim, e_dim, reserved;
} mkgnn_shard_view;
size_t mkgnn_collate_compact_bytes(const int64_t shape[6], int64_t n_molecules, int32_t pad_molecules, int32_t x_dim,
                                   int32_t p_dim, int32_t e_dim);
int mkgnn_collate_compact(const mkgnn_shard_view* shard, int64_t m0, int64_t m1, const int64_t shape[6],
                          int32_t pad_molecules, void* out, size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MOLKGNN_HIP_H */
