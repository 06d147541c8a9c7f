"""torch.autograd bindings of the C ABI: the differentiable operators the
``KernelConv`` / ``KernelSetConv`` / ``MolGCN`` modules are made of.

Every operator launches hand-written gfx950 kernels through
``libmolkgnn_hip.so`` on the current HIP stream; tensors are only used for
device memory.  There is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import torch

from . import _lib
from .plan import BatchPlan

PARAMS_PER_DEGREE = 7   # x_center, x_support, edge_attr_support, p_support, support/center/edge score weights
VARIANTS = {"auto": 0, "generic": 1, "mfma": 2, "bf16": 3}
# mkgnn_kernelsetconv_backward's own switch: "generic" = the one-wave-per-atom kernels + plain gather for every degree,
# "fast" = the MFMA rows / LDS bank / pipelined gather kernels or an error.  None follows the forward variant
# ("generic" -> "generic", anything else -> "auto").
BACKWARD_VARIANTS = {"auto": 0, "generic": 1, "fast": 2}


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _row_major(x: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view with unit inner stride (row stride may exceed the width)."""
    if x.dim() != 2 or x.dtype != torch.float32:
        raise TypeError(f"expected a 2-D float32 tensor, got {tuple(x.shape)} {x.dtype}")
    if x.shape[0] > 1 and x.stride(1) != 1:
        return x.contiguous()
    if x.shape[0] <= 1 and x.stride(-1) != 1:
        return x.contiguous()
    return x


def _aligned_rows(x: torch.Tensor) -> torch.Tensor:
    """Row-major view whose rows start on 16-byte boundaries (row stride a multiple of 4 floats, zero
    padded), which is what the MFMA kernels load with 128-bit accesses.  A no-op for tensors that are
    already laid out that way (everything MolGCN produces); otherwise one padded copy."""
    x = _row_major(x)
    if x.shape[0] == 0 or (_stride0(x) % 4 == 0 and x.data_ptr() % 16 == 0):
        return x
    n, f = x.shape
    store = torch.zeros((n, f + (-f) % 4), dtype=x.dtype, device=x.device)
    store[:, :f] = x
    return store[:, :f]


def _stride0(x: torch.Tensor) -> int:
    return x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1])


def _banks(params: Sequence[torch.Tensor], F: int, E: int):
    banks = _lib.Banks4()
    Ls = []
    keep = []
    for i in range(4):
        xc, xs, es, ps, ts, tc, te = params[i * PARAMS_PER_DEGREE:(i + 1) * PARAMS_PER_DEGREE]
        d = i + 1
        L = int(xc.shape[0])
        if L and (tuple(xs.shape) != (L, d, F) or tuple(xc.shape) != (L, F) or tuple(es.shape) != (L, d, E)):
            raise ValueError(f"degree {d}: kernel shapes {tuple(xc.shape)} {tuple(xs.shape)} {tuple(es.shape)} "
                             f"do not match F={F}, E={E}")
        xc, xs, es, ps = _f32c(xc), _f32c(xs), _f32c(es), _f32c(ps)
        keep += [xc, xs, es, ps]
        b = banks[i]
        b.num_kernels = L
        b.x_center, b.x_support, b.edge_attr_support = _lib.ptr(xc), _lib.ptr(xs), _lib.ptr(es)
        b.p_support = _lib.ptr(ps) if ps.shape[-1] == 3 else None
        b.support_attr_sc_weight, b.center_attr_sc_weight, b.edge_attr_support_sc_weight = \
            ts.data_ptr(), tc.data_ptr(), te.data_ptr()
        Ls.append(L)
    return banks, Ls, keep


def _buckets(plan: BatchPlan, E: int, need_p: bool):
    bk = _lib.Buckets4()
    for i, b in enumerate(plan.buckets):
        k = bk[i]
        k.count = b.count
        if b.count:
            if b.e_nei.numel() != b.count * b.degree * E:
                raise ValueError(f"nei_edge_attr_deg{b.degree} has {b.e_nei.numel()} values, expected "
                                 f"{b.count}x{b.degree}x{E}")
            k.selected_index, k.nei_index, k.nei_edge_attr = b.sel.data_ptr(), b.nei.data_ptr(), b.e_nei.data_ptr()
            k.nei_edge_unit = _lib.ptr(b.e_unit(E))
            if need_p and b.degree == 4:
                if b.p_focal is None or b.p_focal.shape[-1] != 3:
                    raise ValueError("chirality (degree 4, last layer) needs 3-D coordinates")
                k.p_focal, k.nei_p = b.p_focal.data_ptr(), b.nei_p.data_ptr()
    return bk


def workspace_bytes(Ls: Sequence[int], F: int, E: int, n_atoms: int, n_slots: int) -> int:
    return int(_lib.load().mkgnn_workspace_bytes(_lib.Int32x4(*Ls), F, E, n_atoms, n_slots))


def row_inv_norm(x: torch.Tensor) -> torch.Tensor:
    """``1 / max(||x_n||, 1e-8)`` per row (torch.nn.CosineSimilarity's clamp, kernels.py:189)."""
    _lib.require_gpu_tensor(x, "x")
    x = _row_major(x)
    inv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mkgnn_row_inv_norm(x.data_ptr(), _stride0(x), x.shape[0], x.shape[1],
                                                  inv.data_ptr(), _lib.stream_ptr(x.device)), "mkgnn_row_inv_norm")
    return inv


BANK_PREPARED = 0x200        # MKGNN_VARIANT_BANK_PREPARED
ROWS_SPLIT = 0x400           # MKGNN_VARIANT_ROWS_SPLIT / MKGNN_BACKWARD_ROWS_SPLIT
# Pre-split rows (round 6; csrc/kgnn_split.h): a tensor that only the next kernel convolution reads -- h = propagate(sim_sc)
# between two layers (reference KernelLayer.py:119-123 -> kernels.py:527,543), the batch norm's output in front of the
# first -- is written by its producer as the fp16 hi | lo halves the matrix instructions take (same bytes, same strides), so
# that no wave converts a row again.  Such a tensor carries this attribute; its float32 VALUES are meaningless to anybody
# but `kernelsetconv`, which is why only MolGCN.forward / MolKGNNNet.forward ask for it, for tensors they never hand out.
# MKGNN_ROWS_SPLIT=0 turns it off (the library answers mkgnn_rows_split_supported with 0).
_SPLIT_ATTR = "_mkgnn_rows_split"


_PRODUCTS_EPOCH = 0          # bumped by debug_set_products: answers remembered per batch (kernels._accepts_split_rows) go stale


def debug_set_products(forward: int = -1, backward: int = -1) -> None:
    """Diagnostics / bench.py's exact-fp32 leg: 1 / 0 = split-fp16 / fp32 matrix instructions for the node-feature products of
    the streamed forward and backward kernels from the next launch on, -1 = what the environment says
    (``mkgnn_debug_set_forward_products`` / ``_backward_products``).  Pre-split rows follow: they exist only with both on."""
    global _PRODUCTS_EPOCH
    lib = _lib.load()
    _lib.check(lib.mkgnn_debug_set_forward_products(int(forward)), "mkgnn_debug_set_forward_products")
    _lib.check(lib.mkgnn_debug_set_backward_products(int(backward)), "mkgnn_debug_set_backward_products")
    _PRODUCTS_EPOCH += 1


def is_rows_split(x: torch.Tensor) -> bool:
    """``x`` holds pre-split rows (and has not been modified since its producer wrote them)."""
    tag = getattr(x, _SPLIT_ATTR, None)
    return tag is not None and tag == x._version


def mark_rows_split(x: torch.Tensor) -> torch.Tensor:
    setattr(x, _SPLIT_ATTR, x._version)
    return x


def presplit_rows(x: torch.Tensor) -> torch.Tensor:
    """``x`` rewritten as pre-split rows (``mkgnn_rows_presplit``), with its row norms attached: what ``kernelsetconv`` receives
    from the layer before it inside a model.  For benchmarks and tests that time or check a single layer on the operand form
    the training step feeds it; the result is meaningless as float32 values."""
    _lib.require_gpu_tensor(x, "x")
    x = _aligned_rows(x)
    n, F = x.shape
    F4 = F + (-F) % 4
    out = torch.zeros((n, F4), dtype=torch.float32, device=x.device)[:, :F]
    inv = torch.empty(n, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mkgnn_rows_presplit(x.data_ptr(), _stride0(x), n, F, inv.data_ptr(), out.data_ptr(), F4,
                                                   _lib.stream_ptr(x.device)), "mkgnn_rows_presplit")
    setattr(out, _INV_ATTR, (inv, out._version))
    return mark_rows_split(out)


def rows_split_supported(plan: BatchPlan, params: Sequence[torch.Tensor], F: int, E: int, n_atoms: int) -> bool:
    """``mkgnn_rows_split_supported``: would a ``kernelsetconv`` over these banks and buckets, on rows of ``F`` floats in
    16-byte aligned storage, take pre-split rows in its forward and its backward?"""
    banks, Ls, keep = _banks([p.detach() for p in params], F, E)
    buckets = _buckets(plan, E, False)
    F4, K4 = F + (-F) % 4, sum(Ls) + (-sum(Ls)) % 4
    return bool(_lib.load().mkgnn_rows_split_supported(banks, buckets, F4, K4, n_atoms, F, E))


class PreparedBank:
    """The workspace of one forward call with its normalised kernel bank already in it (``prepare_banks``)."""
    __slots__ = ("ws", "key")

    def __init__(self, ws, key):
        self.ws, self.key = ws, key


def _prepared_key(params, F, E, n_atoms, n_slots):
    return (F, E, n_atoms, n_slots) + tuple((p.data_ptr(), p._version) for p in params)


TOUCH = os.environ.get("MKGNN_TOUCH", "1") != "0"      # the batch's index arrays are read once ahead of the first convolution


def plan_touch_list(plan: BatchPlan) -> List[torch.Tensor]:
    """The index arrays of a batch that its convolutions gather through -- per degree ``selected_index``, ``nei_index`` and the unit
    bond rows, and the CSR ``propagate`` sums over (at most 16 arrays; whatever has not been built yet is left out, nothing is
    built for this)."""
    out = []
    for bk in plan.buckets:
        if bk.count:
            out += [bk.nei, bk.sel]
            if bk._e_unit is not None:
                out.append(bk._e_unit)
    if plan._csr_in_packed is not None:
        out += [t for t in plan._csr_in_packed if t is not None]
    return [t for t in out if t.is_cuda and t.numel() * t.element_size() >= 4096][:16]


class touch_hint:
    """``with touch_hint(plan_touch_list(plan)) as h: <batch norm / prepare_banks>`` -- ``mkgnn_touch_hint``: the first of the
    library's launches inside the block that has spare blocks (the batch norm's statistics, else the bank preparation) reads
    these arrays once and discards them, so that the step's first convolution does not find them cold.  Leaving
    the block withdraws a hint nobody took; ``h.taken`` tells afterwards.  The arrays belong to the batch: alive as long as it."""

    def __init__(self, tensors):
        self.tensors = [t for t in (tensors or []) if t.is_cuda][:16] if TOUCH else []
        self.taken = False

    def __enter__(self):
        if self.tensors:
            import ctypes as C
            n = len(self.tensors)
            ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in self.tensors])
            nbytes = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in self.tensors])
            _lib.load().mkgnn_touch_hint(C.cast(ptrs, C.c_void_p), C.cast(nbytes, C.c_void_p), n)
        return self

    def __exit__(self, *exc):
        if self.tensors:
            self.taken = _lib.load().mkgnn_touch_hint(None, None, 0) == 0
        return False


def prepare_banks(params_per_call: Sequence[Sequence[torch.Tensor]], Fs: Sequence[int], E: int, n_atoms: int, n_slots: int,
                  side: Optional["torch.cuda.Stream"] = None, defer: bool = False):
    """``mkgnn_bank_prepare``: the normalised kernel banks of several forward calls (the layers of a model; call k will
    be ``kernelsetconv(x [n_atoms, Fs[k]], ..., params_per_call[k], E, prepared=result[k])``) in ONE launch on the current
    stream -- the banks depend on the parameters only, and one small dependent launch per layer leaves the step.
    Returns one ``PreparedBank`` per call.  ``side``: launch on that stream instead (forked from the current one; buffers are
    still allocated on the current stream) -- the caller joins it, ``cur.wait_stream(side)``, before the first convolution: the
    banks depend on nothing but the parameters, so the launch runs beside whatever precedes the convolutions (the batch norm).
    (Inside a ``touch_hint`` block that nothing has taken yet, the launch's spare blocks read the hinted arrays.)
    ``defer`` (round 6, ``mkgnn_bank_prepare_deferred``; at most 4 calls): nothing is launched -- the preparation is left pending for
    the batch norm's statistics launch that follows (it carries the tasks in blocks of its own), or for ``prepare_flush``; the caller
    MUST call ``prepare_flush`` before the first convolution reads a prepared bank (``prepare_withdraw`` on an error path)."""
    if defer and len(params_per_call) > 4:
        defer = False
    lib = _lib.load()
    dev = params_per_call[0][0].device
    out: List[PreparedBank] = []
    with torch.no_grad():
        for lo in range(0, len(params_per_call), 4):             # (<= 4 calls per launch)
            chunk = params_per_call[lo:lo + 4]
            cnt = len(chunk)
            banks_all = (_lib.KernelBank * (4 * cnt))()
            keep, wss, nbytes = [], [], []
            for k, params in enumerate(chunk):
                banks, Ls, kp = _banks([p.detach() for p in params], Fs[lo + k], E)
                keep += kp
                for i in range(4):
                    banks_all[4 * k + i] = banks[i]
                nb = workspace_bytes(Ls, Fs[lo + k], E, n_atoms, n_slots)
                wss.append(torch.empty(nb, dtype=torch.uint8, device=dev))
                nbytes.append(nb)
            import ctypes as C
            Farr = (C.c_int32 * cnt)(*[int(f) for f in Fs[lo:lo + cnt]])
            wsarr = (C.c_void_p * cnt)(*[w.data_ptr() for w in wss])
            nbarr = (C.c_size_t * cnt)(*nbytes)
            with torch.cuda.device(dev):
                cur = torch.cuda.current_stream(dev)
                if side is not None and side != cur:
                    side.wait_stream(cur)
                if defer:
                    _lib.check(lib.mkgnn_bank_prepare_deferred(cnt, C.cast(banks_all, C.c_void_p), C.cast(Farr, C.c_void_p), E,
                                                               C.cast(wsarr, C.c_void_p), C.cast(nbarr, C.c_void_p)),
                               "mkgnn_bank_prepare_deferred")
                    _PREPARE_HELD[:] = [keep, wss]       # (read at the launch: alive until the flush)
                    for k, params in enumerate(chunk):
                        out.append(PreparedBank(wss[k], _prepared_key(params, Fs[lo + k], E, n_atoms, n_slots)))
                    continue
                with torch.cuda.stream(side if side is not None else cur):
                    _lib.check(lib.mkgnn_bank_prepare(cnt, C.cast(banks_all, C.c_void_p), C.cast(Farr, C.c_void_p), E,
                                                      C.cast(wsarr, C.c_void_p), C.cast(nbarr, C.c_void_p), _lib.stream_ptr(dev)),
                               "mkgnn_bank_prepare")
            for k, params in enumerate(chunk):
                out.append(PreparedBank(wss[k], _prepared_key(params, Fs[lo + k], E, n_atoms, n_slots)))
    return out


_PREPARE_HELD: list = []     # what a pending (deferred) bank preparation will read


def prepare_flush(device) -> None:
    """``mkgnn_bank_prepare_flush``: a deferred bank preparation nobody has carried yet is launched on the current stream (no-op:
    none pending)."""
    try:
        with torch.cuda.device(device):
            _lib.check(_lib.load().mkgnn_bank_prepare_flush(_lib.stream_ptr(device)), "mkgnn_bank_prepare_flush")
    finally:
        _PREPARE_HELD.clear()


def prepare_withdraw(device) -> bool:
    """``mkgnn_bank_prepare_withdraw``: drop a pending preparation (an error path: its workspaces are about to go)."""
    try:
        with torch.cuda.device(device):
            return bool(_lib.load().mkgnn_bank_prepare_withdraw())
    finally:
        _PREPARE_HELD.clear()


def _forward_impl(x, plan: BatchPlan, is_last_layer: bool, variant: int, out_pad: int, E: int, params, want_saved: bool,
                  inv=None, prepared: Optional[PreparedBank] = None):
    lib = _lib.load()
    _lib.require_gpu_tensor(x, "x")
    x_split = is_rows_split(x)
    if x_split:
        if _stride0(x) % 4 or x.data_ptr() % 16 or x.stride(1) != 1 or inv is None:
            raise _lib.MolKGNNLibraryError("pre-split rows must stay in the storage their producer wrote (with their row norms)")
        variant |= ROWS_SPLIT
    else:
        x = _row_major(x) if (variant & 0xFF) == VARIANTS["generic"] else _aligned_rows(x)
    n, F = x.shape
    dev = x.device
    banks, Ls, keep = _banks(params, F, E)
    K = sum(Ls)
    skipped = any(b.count and Ls[i] == 0 for i, b in enumerate(plan.buckets))
    need_p = bool(is_last_layer) and plan.buckets[3].count > 0
    buckets = _buckets(plan, E, need_p)
    if out_pad is None:                      # default: storage rows padded to 16 bytes, the result is the [:, :K] view
        out_pad = (-K) % 4
    out_w = K + out_pad
    if out_pad and (out_w != (K + 3) // 4 * 4):
        raise ValueError("out_pad must round the output width up to a multiple of 4 (the library zeroes exactly that padding)")
    out_full = torch.empty((n, out_w), dtype=torch.float32, device=dev)      # the forward call zeroes it (and the padding)
    saved = _lib.Saved4()
    saved_t = []
    for i, b in enumerate(plan.buckets):
        L = Ls[i]
        if want_saved and b.count and L:
            # one 16-byte record per (atom, kernel) pair: support, centre, edge score, chosen permutation (int32 bits)
            pr = torch.empty((b.count, L, 4), dtype=torch.float32, device=dev)
            ch = torch.empty((b.count, L), dtype=torch.int8, device=dev) if (i == 3 and is_last_layer) else None
            saved[i].pair_state = pr.data_ptr()
            saved[i].chirality = ch.data_ptr() if ch is not None else None
            saved_t.append((pr, ch))
        else:
            saved_t.append((None, None))
    with torch.cuda.device(dev):
        st = _lib.stream_ptr(dev)
        if inv is None:                      # the producer of x did not hand its row norms over
            inv = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(lib.mkgnn_row_inv_norm(x.data_ptr(), _stride0(x), n, F, inv.data_ptr(), st), "mkgnn_row_inv_norm")
        ws_bytes = workspace_bytes(Ls, F, E, n, plan.n_slots)
        if prepared is not None and prepared.key == _prepared_key(params, F, E, n, plan.n_slots) \
                and prepared.ws.numel() >= ws_bytes:
            ws = prepared.ws                  # the bank is in it already (prepare_banks)
            variant |= BANK_PREPARED
        else:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if _USE_TORCH_OPS:
            ps, bkl, counts, sv = _op_lists(plan, params, E, need_p, saved_t, dev)
            _lib.load_torch_ops().kernelsetconv_forward(x, inv, ps, bkl, counts, E, bool(is_last_layer), out_full, sv, ws, variant)
        else:
            _lib.check(lib.mkgnn_kernelsetconv_forward(
                banks, buckets, x.data_ptr(), _stride0(x), inv.data_ptr(), n, F, E, int(bool(is_last_layer)),
                out_full.data_ptr(), out_w, saved, ws.data_ptr(), ws_bytes, variant, st),
                "mkgnn_kernelsetconv_forward")
    return x, out_full, inv, saved_t, Ls, ws, x_split


# MKGNN_TORCH_OPS=1: the forward / backward calls go through the registered operators (torch.ops.molkgnn.*,
# csrc/torch_ops.cpp: TORCH_LIBRARY over the C ABI) instead of ctypes -- same kernels, same results (GPU test)
_USE_TORCH_OPS = os.environ.get("MKGNN_TORCH_OPS") == "1"


def use_torch_ops(on: bool) -> None:
    """Route ``kernelsetconv``'s forward / backward calls through ``torch.ops.molkgnn`` (True) or ctypes (False)."""
    global _USE_TORCH_OPS
    if on:
        _lib.load_torch_ops()
    _USE_TORCH_OPS = bool(on)


def _op_lists(plan: BatchPlan, params, E: int, need_p: bool, saved_t, dev):
    """The flat tensor lists of the registered operators (torch_ops.cpp): 28 parameters, 24 bucket tensors, 4 counts, 8 saved."""
    none = torch.empty(0, dtype=torch.float32, device=dev)
    ps = []
    for i in range(4):
        xc, xs, es, p3, ts, tc, te = params[i * PARAMS_PER_DEGREE:(i + 1) * PARAMS_PER_DEGREE]
        ps += [_f32c(xc), _f32c(xs), _f32c(es), _f32c(p3), ts.detach(), tc.detach(), te.detach()]
    bk, counts = [], []
    for b in plan.buckets:
        counts.append(int(b.count))
        if not b.count:
            bk += [none] * 6
            continue
        if b.e_nei.numel() != b.count * b.degree * E:
            raise ValueError(f"nei_edge_attr_deg{b.degree} has {b.e_nei.numel()} values, expected {b.count}x{b.degree}x{E}")
        eu = b.e_unit(E)
        withp = need_p and b.degree == 4
        if withp and (b.p_focal is None or b.p_focal.shape[-1] != 3):
            raise ValueError("chirality (degree 4, last layer) needs 3-D coordinates")
        bk += [b.sel, b.nei, b.e_nei, b.p_focal if withp else none, b.nei_p if withp else none, eu if eu is not None else none]
    sv = []
    for pr, ch in saved_t:
        sv += [pr if pr is not None else none, ch if ch is not None else none]
    return ps, bk, counts, sv


def kernelsetconv_details(x, plan: BatchPlan, is_last_layer: bool, params, edge_attr_dim: int, variant: str = "auto",
                          raw: bool = False):
    """Forward only, returning what the kernels keep for backward as well:
    ``(out [N, K], [(best_index [L_d, N_d] uint8, scores [3, L_d, N_d], chirality [L_d, N_d]) per degree])``
    (unpacked from the atom-major pair records of ``mkgnn_saved``; ``raw=True`` returns the records themselves,
    ``(pair_state [N_d, L_d, 4], chirality [N_d, L_d])``, without any further device work).  Used by the parity tests for
    the tie-aware criterion and by bench.py's forward timing."""
    with torch.no_grad():
        _, out, _, saved_t, _, _, _ = _forward_impl(x, plan, is_last_layer, VARIANTS[variant], 0, edge_attr_dim,
                                                 [p.detach() for p in params], True, _handed_inv_norm(x))
    if raw:
        return out, saved_t

    def unpack(pr, ch):
        if pr is None:
            return None, None, None
        best = pr[..., 3].contiguous().view(torch.int32).to(torch.uint8).t()
        return best, pr[..., :3].permute(2, 1, 0), (None if ch is None else ch.t())
    return out, [unpack(pr, ch) for pr, ch in saved_t]


DEFER_BANK = 0x100           # MKGNN_BACKWARD_DEFER_BANK
THROUGH_NEIGHBOURS = 0x200   # MKGNN_BACKWARD_THROUGH_NEIGHBOURS


class _Deferred:
    """State of an open ``deferred_bank_gradients`` region (one per process: backward passes do not nest)."""
    active = False
    held: list = []          # everything the unjoined helper-stream kernels read or write
    seen: set = set()        # ids of the parameters whose gradients are still in flight


class deferred_bank_gradients:
    """``with deferred_bank_gradients(): loss.backward()`` -- inside, a KernelSetConv backward hands its input gradient
    to the layer below as soon as the x-gradient chain is done and leaves its kernel-bank gradients running on the
    library's helper stream (MKGNN_BACKWARD_DEFER_BANK); leaving the region makes the current stream wait for all of them
    (mkgnn_backward_join).  Only the optimiser reads a weight gradient, so the bank chain of layer l overlaps the whole
    backward of layer l - 1 instead of holding it up.

    The gradients of the parameters are undefined until the region is left: a call whose parameters already have a
    ``.grad`` (it would be added to at once), or that shares a parameter with an earlier call of the region, is not
    deferred.  Buffers the helper kernels use stay referenced until the join, so the allocator cannot hand them out."""

    def __enter__(self):
        if _Deferred.active:
            raise RuntimeError("deferred_bank_gradients regions do not nest")
        _Deferred.active, _Deferred.held, _Deferred.seen = True, [], set()
        return self

    def __exit__(self, *exc):
        _Deferred.active = False
        try:
            if _Deferred.held and torch.cuda.is_available():
                _lib.check(_lib.load().mkgnn_backward_join(_lib.stream_ptr(_Deferred.held[0][0].device)), "mkgnn_backward_join")
        finally:
            _Deferred.held, _Deferred.seen = [], set()
        return False


class _KernelSetConvFn(torch.autograd.Function):
    """BaseKernelSetConv.forward (reference kernels.py:610-751) as one differentiable operator."""

    @staticmethod
    def forward(ctx, x, plan: BatchPlan, is_last_layer: bool, variant: int, out_pad: int, E: int, inv, bwd_variant: int,
                propagate, prepared, *params):
        need_grad = any(ctx.needs_input_grad)
        ctx.bwd_variant = int(bwd_variant)
        x, out_full, inv, saved_t, Ls, ws, x_split = _forward_impl(x, plan, is_last_layer, variant, out_pad, E, params, need_grad,
                                                                   inv, prepared)
        ctx.x_split = x_split
        ctx.plan, ctx.is_last, ctx.E, ctx.Ls = plan, bool(is_last_layer), E, Ls
        # the workspace holds the normalised kernel bank: backward reuses it (and the buffer) instead of redoing it
        ctx.ws = ws if need_grad else None
        ctx.param_versions = [p._version for p in params]
        ctx.saved_t = saved_t
        ctx.save_for_backward(x, inv, *params)
        ctx.propagate = bool(propagate)
        K = sum(Ls)
        sim_sc = out_full[:, :K] if out_full.shape[1] != K else out_full
        if not propagate:
            return sim_sc
        # ... followed by MolGCN.propagate (KernelLayer.py:119-123) on the block rows, as ONE differentiable operator: its
        # backward hands the gradient of h straight to the kernels (MKGNN_BACKWARD_THROUGH_NEIGHBOURS) where they can
        # fold the propagate step's gradient in, instead of making a pass over the edges for it
        # (propagate == "split": h is written pre-split -- mode 3 -- for a caller whose only reader is the next convolution)
        inv_h = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        h = _segment_sum_blocks(sim_sc, plan.csr_in_packed, None, tuple(Ls), 3 if propagate == "split" else 1, (-K) % 4, inv_h)
        ctx.mark_non_differentiable(inv_h)
        ctx.set_materialize_grads(False)
        return h, inv_h

    @staticmethod
    def backward(ctx, grad_out, _grad_inv=None):
        lib = _lib.load()
        x, inv, *params = ctx.saved_tensors
        plan, E, Ls = ctx.plan, ctx.E, ctx.Ls
        n, F = x.shape
        dev = x.device
        if grad_out is None:
            return (None,) * (10 + len(params))
        g = _row_major(grad_out if grad_out.dtype == torch.float32 else grad_out.float())
        banks, _, keep = _banks(params, F, E)
        buckets = _buckets(plan, E, False)
        through = 0
        if ctx.propagate:
            # g is d loss / d h.  The streamed kernels sum the neighbours' rows of it themselves; anything else gets the
            # propagate step's gradient (every atom's own block of d sim_sc) from its own pass first
            with torch.cuda.device(dev):
                streams = ctx.bwd_variant != BACKWARD_VARIANTS["generic"] and \
                    lib.mkgnn_backward_streams(banks, buckets, x.data_ptr(), _stride0(x), n, F, E) == 1
            if streams and not os.environ.get("MKGNN_NO_THROUGH_NEIGHBOURS"):
                through = THROUGH_NEIGHBOURS
            else:
                K = sum(Ls)
                if _stride0(g) % 4 or g.data_ptr() % 16:
                    g = _aligned_rows(g)
                g = _segment_sum_blocks(g, plan.csr_out, plan.deg8, tuple(Ls), 2, (-K) % 4, None)
        saved = _lib.Saved4()
        for i, (pr, ch) in enumerate(ctx.saved_t):
            saved[i].pair_state = _lib.ptr(pr)
            saved[i].chirality = _lib.ptr(ch)
        grads = _lib.BankGrads4()
        gparams: List[Optional[torch.Tensor]] = []
        alive = []          # every buffer whose address goes to the library stays referenced until the call is enqueued
        for i in range(4):
            xc, xs, es, ps, ts, tc, te = params[i * PARAMS_PER_DEGREE:(i + 1) * PARAMS_PER_DEGREE]
            gxc, gxs, ges = torch.empty_like(xc), torch.empty_like(xs), torch.empty_like(es)
            # a degree without atoms, or without kernels in this set (a fixed / trainable split), launches nothing
            # that writes the score-weight partials: start from zeros there, the values are dropped below anyway
            idle = plan.buckets[i].count == 0 or Ls[i] == 0
            gth = torch.zeros(3, dtype=torch.float32, device=dev) if idle else torch.empty(3, dtype=torch.float32, device=dev)
            alive += [gxc, gxs, ges, gth]     # (an absent degree's buffers are written too -- with zeros -- and dropped afterwards)
            gr = grads[i]
            gr.x_center, gr.x_support, gr.edge_attr_support = _lib.ptr(gxc), _lib.ptr(gxs), _lib.ptr(ges)
            gr.support_attr_sc_weight = gth.data_ptr()
            gr.center_attr_sc_weight = gth.data_ptr() + 4
            gr.edge_attr_support_sc_weight = gth.data_ptr() + 8
            # p_support is not differentiable in the reference (kernels.py:279-350): gradient stays None
            if idle:
                # no atom of this degree in the batch, or no kernels of this degree in this set (the score weights
                # handed in for such a degree belong to ANOTHER degree's KernelConv, kernels._bank_params): the
                # reference's autograd leaves all of these gradients None
                gparams += [None] * PARAMS_PER_DEGREE
            else:
                gparams += [gxc, gxs, ges, None, gth[0].reshape(ts.shape), gth[1].reshape(tc.shape),
                            gth[2].reshape(te.shape)]
        rowptr, rows = plan.scatter
        with torch.cuda.device(dev):
            # 16-byte rows: the gather and whatever consumes the gradient next use 128-bit accesses
            F4 = F + (-F) % 4
            gx = torch.empty((n, F4), dtype=torch.float32, device=dev)[:, :F] if ctx.needs_input_grad[0] else None
            ws_bytes = workspace_bytes(Ls, F, E, n, plan.n_slots)
            ws, ctx.ws = ctx.ws, None
            reuse = ws is not None and ws.numel() >= ws_bytes and \
                all(p._version == v for p, v in zip(params, ctx.param_versions))
            if not reuse:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            bwd_variant = ctx.bwd_variant | through | (ROWS_SPLIT if ctx.x_split else 0)
            live = [p for p in params if p.requires_grad]
            defer = _Deferred.active and all(p.grad is None and id(p) not in _Deferred.seen for p in live)
            if defer:
                bwd_variant |= DEFER_BANK
                _Deferred.seen.update(id(p) for p in live)
                # (NOT the gradient tensors themselves: autograd takes a returned gradient over as .grad only while
                # nobody else refers to it, and copies it -- reading it too early -- otherwise; their memory stays put
                # because .grad holds it)
                _Deferred.held.append((x, inv, g, ws, ctx.saved_t, plan, params, [t for t in alive if t is not None and
                                                                                 all(t is not q for q in gparams)]))
            if _USE_TORCH_OPS:
                ps, bkl, counts, sv = _op_lists(plan, params, E, False, ctx.saved_t, dev)
                _lib.load_torch_ops().kernelsetconv_backward(x, inv, ps, bkl, counts, E, bool(ctx.is_last), g, sv, rowptr, rows, gx,
                                                             list(alive), ws, bool(reuse), bwd_variant)
            else:
                _lib.check(lib.mkgnn_kernelsetconv_backward(
                    banks, buckets, x.data_ptr(), _stride0(x), inv.data_ptr(), n, F, E, int(ctx.is_last),
                    g.data_ptr(), _stride0(g), saved, rowptr.data_ptr(), rows.data_ptr(),
                    _lib.ptr(gx), F4, grads, ws.data_ptr(), ws_bytes, int(reuse), bwd_variant, _lib.stream_ptr(dev)),
                    "mkgnn_kernelsetconv_backward")
        del alive           # (freed memory is only handed out again in stream order, after the kernels above)
        return (gx, None, None, None, None, None, None, None, None, None, *gparams)


BLOCK_ROWS = 0x100           # MKGNN_VARIANT_BLOCK_ROWS
_BLOCKS_ATTR = "_mkgnn_block_rows"


def kernelsetconv(x: torch.Tensor, plan: BatchPlan, is_last_layer: bool, params: Sequence[torch.Tensor],
                  edge_attr_dim: int, variant: str = "auto", out_pad: Optional[int] = None,
                  block_rows: bool = False, backward_variant: Optional[str] = None, propagate: bool = False,
                  prepared: Optional[PreparedBank] = None) -> torch.Tensor:
    """``[N, F] -> [N, K]`` kernel convolution over the four degree buckets of ``plan``.

    ``params`` is the flat list, degree 1..4, of (x_center, x_support,
    edge_attr_support, p_support, support_attr_sc_weight, center_attr_sc_weight,
    edge_attr_support_sc_weight).  The storage rows are padded with zero columns
    to a multiple of 4 floats (``out_pad=None``; the returned tensor is the
    ``[:, :K]`` view) so that whatever reads the result next gets 16-byte rows;
    ``out_pad=0`` gives contiguous storage.

    ``propagate=True`` (with ``block_rows=True``) returns ``h = propagate_add(sim_sc)`` instead of ``sim_sc``: convolution and
    neighbour sum as one operator, whose backward skips the propagate step's own gradient pass where the kernels can fold
    it in.  ``block_rows=True`` is for a caller that hands the result straight to ``propagate_add`` (as ``MolGCN.forward``
    does): only every atom's own column block is written -- the zeros elsewhere are implied, nothing reads them --
    and the gradient it gets back is defined only there.  The tensor carries the block sizes for ``propagate_add``.
    """
    if block_rows and out_pad not in (None, (-sum(int(p.shape[0]) for p in params[0::7])) % 4):
        raise ValueError("block_rows needs the default padded storage")
    if backward_variant is None:
        backward_variant = "generic" if variant == "generic" else "auto"
    if propagate:
        if not block_rows:
            raise ValueError("propagate=True continues on block rows: block_rows=True is required")
        if propagate not in (True, "split"):
            raise ValueError('propagate: False, True, or "split" (h written pre-split for the next kernelsetconv)')
        h, inv_h = _KernelSetConvFn.apply(x, plan, is_last_layer, VARIANTS[variant] | BLOCK_ROWS, out_pad, edge_attr_dim,
                                          _handed_inv_norm(x), BACKWARD_VARIANTS[backward_variant], propagate, prepared, *params)
        setattr(h, _INV_ATTR, (inv_h, h._version))
        if propagate == "split":
            mark_rows_split(h)
        return h
    out = _KernelSetConvFn.apply(x, plan, is_last_layer, VARIANTS[variant] | (BLOCK_ROWS if block_rows else 0), out_pad,
                                 edge_attr_dim, _handed_inv_norm(x), BACKWARD_VARIANTS[backward_variant], False, prepared,
                                 *params)
    if block_rows:
        setattr(out, _BLOCKS_ATTR, tuple(int(p.shape[0]) for p in params[0::7]))
    return out


_INV_ATTR = "_mkgnn_inv_norm"


def _handed_inv_norm(x: torch.Tensor):
    """Row norms a previous operator attached to ``x`` (see propagate_add), if ``x`` is still that tensor's data."""
    tag = getattr(x, _INV_ATTR, None)
    if tag is None:
        return None
    inv, version = tag
    if version != x._version or inv.shape[0] != x.shape[0] or inv.device != x.device:
        return None                               # modified in place since: recompute
    return inv


class _SegmentSumFn(torch.autograd.Function):
    """``out[i] = sum_{j -> i} v[j]`` over ``edge_index`` (MolGCN.propagate, aggr='add').  ``blocks`` (the four
    per-degree kernel counts) marks ``v`` as block rows: see ``mkgnn_segment_sum_block_rows``."""

    @staticmethod
    def forward(ctx, v, plan: BatchPlan, out_pad: int, blocks):
        _lib.require_gpu_tensor(v, "sim_sc")
        v = _row_major(v)
        ctx.plan = plan
        ctx.width = v.shape[1]
        ctx.blocks = blocks
        inv = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
        if blocks is not None:
            out = _segment_sum_blocks(v, plan.csr_in_packed, None, blocks, 1, out_pad, inv)
        else:
            out = _segment_sum(v, plan.csr_in, out_pad, inv)
        ctx.mark_non_differentiable(inv)
        ctx.set_materialize_grads(False)             # no zero tensor for the norms' (non-existent) gradient
        return out, inv

    @staticmethod
    def backward(ctx, g, _g_inv):
        if g is None:
            return None, None, None, None
        g = _row_major(g if g.dtype == torch.float32 else g.float())
        pad = (-g.shape[1]) % 4
        if ctx.blocks is not None and _stride0(g) % 4 == 0 and g.data_ptr() % 16 == 0:
            # only every atom's own block of d sim_sc is defined: all the convolution's backward reads
            return _segment_sum_blocks(g, ctx.plan.csr_out, ctx.plan.deg8, ctx.blocks, 2, pad, None), None, None, None
        return _segment_sum(g, ctx.plan.csr_out, pad), None, None, None


def _segment_sum_blocks(v, csr, deg8, blocks, mode: int, out_pad: int, inv) -> torch.Tensor:
    rowptr, col = csr
    n, w = v.shape
    if sum(blocks) != w or (w + out_pad) % 4 or _stride0(v) % 4 or v.data_ptr() % 16:
        raise _lib.MolKGNNLibraryError("block-row propagate needs 16-byte aligned rows whose width is the sum of the blocks")
    # mode 1 (and 3: the same rows, pre-split) writes whole rows (alignment padding included); mode 2 leaves everything outside the blocks undefined --
    # the padding too: its only reader is the convolution's backward, which reads blocks
    out = torch.empty((n, w + out_pad), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(_lib.load().mkgnn_segment_sum_block_rows(
            v.data_ptr(), _stride0(v), rowptr.data_ptr(), col.data_ptr(), _lib.ptr(deg8), n, _lib.Int32x4(*blocks), mode,
            out.data_ptr(), w + out_pad, _lib.ptr(inv), _lib.stream_ptr(v.device)), "mkgnn_segment_sum_block_rows")
    return out[:, :w] if out_pad else out


def _segment_sum(v: torch.Tensor, csr, out_pad: int, inv: Optional[torch.Tensor] = None) -> torch.Tensor:
    rowptr, col = csr
    n, w = v.shape
    if out_pad and (w + out_pad) != (w + 3) // 4 * 4:
        raise ValueError("out_pad must round the width up to a multiple of 4")
    # padded storage is written in full by the aligned kernel; other layouts get their padding zeroed here
    aligned = (w + out_pad) % 4 == 0 and _stride0(v) % 4 == 0 and v.data_ptr() % 16 == 0 and w <= 256 and col.numel() > 0
    alloc = torch.zeros if (out_pad and not aligned) else torch.empty
    out = alloc((n, w + out_pad), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(_lib.load().mkgnn_segment_sum_rows(v.data_ptr(), _stride0(v), rowptr.data_ptr(), _lib.ptr(col), n, w,
                                                      out.data_ptr(), w + out_pad, _lib.ptr(inv),
                                                      _lib.stream_ptr(v.device)),
                   "mkgnn_segment_sum_rows")
    return out[:, :w] if out_pad else out


def propagate_add(sim_sc: torch.Tensor, plan: BatchPlan, out_pad: int = 0) -> torch.Tensor:
    """MolGCN.propagate with aggr='add'.  The kernel also emits ``1 / max(||h_n||, 1e-8)``; it rides on the
    returned tensor so that the next kernel convolution does not make another pass over ``h`` for it.
    A ``sim_sc`` made by ``kernelsetconv(..., block_rows=True)`` is summed block by block (5x fewer bytes per edge)."""
    blocks = getattr(sim_sc, _BLOCKS_ATTR, None)
    h, inv = _SegmentSumFn.apply(sim_sc, plan, out_pad, blocks)
    setattr(h, _INV_ATTR, (inv, h._version))
    return h
