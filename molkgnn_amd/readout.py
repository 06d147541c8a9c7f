"""The two consumers either side of the kernel-convolution stack as HIP operators:

* ``batch_norm(x, bn)``  -- ``self.node_batch_norm(data.x)`` (reference ``MolKGNNNet.py:115``) for a
  ``torch.nn.BatchNorm1d`` module ``bn`` (its parameters, running statistics and flags are used as is);
* ``readout(h, lin1, lin2, dropout, batch, size)`` -- ``pool(lin2(dropout(swish(lin1(h)))), batch)``
  (reference ``MolKGNNNet.py:144-146``).

Both call ``libmolkgnn_hip.so`` through the C ABI (``mkgnn_batchnorm_*``, ``mkgnn_readout_*``); shapes
outside the kernels' limits (hidden width > 64, node width > 128, an unsorted ``batch`` vector,
``momentum=None``) take the same formula through PyTorch operators on the GPU instead.
"""
from __future__ import annotations

import ctypes

import weakref
from typing import Optional

import os

import torch

from . import _lib
from .functional import _aligned_rows, _row_major, _stride0

_SEG_CACHE: dict = {}
_SEG_CACHE_MAX = 32


class MoleculeSegments:
    """``batch`` (atom -> molecule id) as contiguous segments: ``mol_ptr [B+1]``, ``atom_mol [N]`` (int32)."""

    @classmethod
    def from_tensors(cls, mol_ptr: torch.Tensor, atom_mol: torch.Tensor, max_atoms: Optional[int] = None,
                     max_edges: Optional[int] = None) -> "MoleculeSegments":
        """Segments handed over by the loader (int32, on the GPU, atoms of a molecule contiguous): no derivation, no
        host synchronisation; the tensors may be refilled in place between replays of a captured step.  ``max_atoms`` /
        ``max_edges``: the loader's bound on a molecule's atoms / directed edges for EVERY batch these tensors will hold
        (``padding.pad_batch`` knows them) -- without it the fused tail (``tail_loss``), which needs the bound, is not taken:
        the contents cannot be inspected without a host synchronisation, and must not decide differently in an eager step
        and in a captured one."""
        if mol_ptr.dtype != torch.int32 or atom_mol.dtype != torch.int32:
            raise TypeError("mol_ptr and atom_mol must be int32")
        seg = cls.__new__(cls)
        seg.size = int(mol_ptr.numel()) - 1
        seg.sorted = True
        seg.mol_ptr, seg.atom_mol = mol_ptr, atom_mol
        seg.from_loader = True
        seg.max_atoms = None if max_atoms is None else int(max_atoms)
        seg.max_edges = None if max_edges is None else int(max_edges)
        return seg

    def __init__(self, batch: torch.Tensor, size: int):
        self.size = int(size)
        b = batch.long()
        self.sorted = bool((b[1:] >= b[:-1]).all().item()) if b.numel() > 1 else True
        counts = torch.bincount(b, minlength=self.size)
        if counts.numel() > self.size:
            raise ValueError(f"batch holds molecule ids >= size ({counts.numel()} > {self.size})")
        ptr = torch.zeros(self.size + 1, dtype=torch.int32, device=batch.device)
        ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        self.mol_ptr = ptr
        self.atom_mol = b.to(torch.int32).contiguous()


def molecule_segments(batch: torch.Tensor, size: Optional[int]) -> MoleculeSegments:
    """Cached on the identity of ``batch`` (resident batches keep their tensors alive): built once per
    batch, so a step contains no host synchronisation for it (hipGraph capture)."""
    if size is None:
        size = int(batch.max().item()) + 1 if batch.numel() else 0
    key = (batch.data_ptr(), batch.numel(), str(batch.device), int(size), batch._version)   # (refilled in place: rebuilt)
    hit = _SEG_CACHE.get(key)
    if hit is not None:
        seg, ref = hit
        if ref() is batch:                       # addresses are recycled: only the very tensor it was built from counts
            return seg
        del _SEG_CACHE[key]
    seg = MoleculeSegments(batch, size)
    if len(_SEG_CACHE) >= _SEG_CACHE_MAX:
        _SEG_CACHE.pop(next(iter(_SEG_CACHE)))
    _SEG_CACHE[key] = (seg, weakref.ref(batch))
    return seg


def swish(x):
    return x * torch.sigmoid(x)


def readout_supported(F: int, H: int, G: int) -> bool:
    return 1 <= F <= 128 and 1 <= H <= 64 and 1 <= G <= 64


def _params(w1, b1, w2, b2):
    p = _lib.ReadoutParams()
    p.lin1_weight, p.lin1_bias = w1.data_ptr(), _lib.ptr(b1)
    p.lin2_weight, p.lin2_bias = w2.data_ptr(), _lib.ptr(b2)
    p.H, p.F = w1.shape
    p.G = w2.shape[0]
    return p


class _ReadoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, w1, b1, w2, b2, keep, seg: MoleculeSegments):
        lib = _lib.load()
        _lib.require_gpu_tensor(h, "node_representation")
        h = _aligned_rows(h if h.dtype == torch.float32 else h.float())
        w1c, w2c = w1.contiguous(), w2.contiguous()
        n, F = h.shape
        H, G = w1c.shape[0], w2c.shape[0]
        dev = h.device
        hs = lib.mkgnn_readout_hidden_stride(H)
        pre = torch.empty((n, hs), dtype=torch.float32, device=dev)
        pooled = torch.empty((seg.size, hs), dtype=torch.float32, device=dev)
        out = torch.empty((seg.size, G), dtype=torch.float32, device=dev)
        keepc = None if keep is None else keep.contiguous()
        p = _params(w1c, b1, w2c, b2)
        with torch.cuda.device(dev):
            _lib.check(lib.mkgnn_readout_forward(p, h.data_ptr(), _stride0(h), n, seg.mol_ptr.data_ptr(), seg.size,
                                                 _lib.ptr(keepc), _lib.ptr(pre), _lib.ptr(pooled), _lib.ptr(out), G,
                                                 _lib.stream_ptr(dev)), "mkgnn_readout_forward")
        ctx.seg = seg
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.save_for_backward(h, w1c, b1, w2c, b2, keepc, pre, pooled)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        h, w1, b1, w2, b2, keep, pre, pooled = ctx.saved_tensors
        seg = ctx.seg
        n, F = h.shape
        H, G = w1.shape[0], w2.shape[0]
        dev = h.device
        g = _row_major(grad_out if grad_out.dtype == torch.float32 else grad_out.float())
        F4 = F + (-F) % 4                       # 16-byte rows for the consumer of the gradient (propagate's backward)
        gh = torch.empty((n, F4), dtype=torch.float32, device=dev)[:, :F] if ctx.needs_input_grad[0] else None
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        gb1 = torch.empty_like(b1) if b1 is not None else None
        gb2 = torch.empty_like(b2) if b2 is not None else None
        if n == 0 or seg.size == 0:
            for t in (gh, gw1, gw2, gb1, gb2):
                if t is not None:
                    t.zero_()
            return gh, gw1, gb1, gw2, gb2, None, None
        p = _params(w1, b1, w2, b2)
        with torch.cuda.device(dev):
            ws_bytes = int(lib.mkgnn_readout_workspace_bytes(F, H, G, n, seg.size))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.mkgnn_readout_backward(
                p, h.data_ptr(), _stride0(h), n, seg.mol_ptr.data_ptr(), seg.atom_mol.data_ptr(), seg.size,
                _lib.ptr(keep), pre.data_ptr(), pooled.data_ptr(), g.data_ptr(), _stride0(g),
                _lib.ptr(gh), F4, gw1.data_ptr(), _lib.ptr(gb1), gw2.data_ptr(), _lib.ptr(gb2),
                ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev)), "mkgnn_readout_backward")
        return gh, gw1, gb1, gw2, gb2, None, None


def _sel_buckets(plan):
    bk = _lib.Buckets4()
    for i, b in enumerate(plan.buckets):
        bk[i].count = b.count
        if b.count:
            bk[i].selected_index = b.sel.data_ptr()
    return bk


class _ReadoutBlocksFn(torch.autograd.Function):
    """``pool(lin2(dropout(swish(lin1(propagate(sim))))))`` from the block rows ``sim`` of the last kernel convolution:
    ``mkgnn_readout_blocks_forward`` / ``_backward`` (projection first, then the propagate step on H-wide rows)."""

    @staticmethod
    def forward(ctx, sim, w1, b1, w2, b2, keep, seg: MoleculeSegments, plan, blocks):
        lib = _lib.load()
        _lib.require_gpu_tensor(sim, "sim_sc")
        n, K = sim.shape
        w1c, w2c = w1.contiguous(), w2.contiguous()
        H, G = w1c.shape[0], w2c.shape[0]
        dev = sim.device
        hs = lib.mkgnn_readout_hidden_stride(H)
        z = torch.empty((n, hs), dtype=torch.float32, device=dev)
        pre = torch.empty((n, hs), dtype=torch.float32, device=dev)
        pooled = torch.empty((seg.size, hs), dtype=torch.float32, device=dev)
        out = torch.empty((seg.size, G), dtype=torch.float32, device=dev)
        # a backward will follow: the forward leaves the gate keep * swish'(pre + b1) in place of pre, and its sums per molecule
        gsum = torch.empty((seg.size, hs), dtype=torch.float32, device=dev) if any(ctx.needs_input_grad) else None
        keepc = None if keep is None else keep.contiguous()
        p = _params(w1c, b1, w2c, b2)
        rowptr, col = plan.csr_in
        with torch.cuda.device(dev):
            _lib.check(lib.mkgnn_readout_blocks_forward(
                p, sim.data_ptr(), _stride0(sim), _lib.Int32x4(*blocks), _sel_buckets(plan), n, rowptr.data_ptr(), col.data_ptr(),
                seg.mol_ptr.data_ptr(), seg.size, _lib.ptr(keepc), z.data_ptr(), pre.data_ptr(), pooled.data_ptr(), _lib.ptr(gsum),
                out.data_ptr(), G, _lib.stream_ptr(dev)), "mkgnn_readout_blocks_forward")
        ctx.seg, ctx.plan, ctx.blocks = seg, plan, tuple(blocks)
        ctx.save_for_backward(sim, w1c, b1, w2c, b2, pre, gsum, pooled)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        sim, w1, b1, w2, b2, gate, gsum, pooled = ctx.saved_tensors
        seg, plan, blocks = ctx.seg, ctx.plan, ctx.blocks
        n, K = sim.shape
        H, G = w1.shape[0], w2.shape[0]
        dev = sim.device
        g = _row_major(grad_out if grad_out.dtype == torch.float32 else grad_out.float())
        hs = gate.shape[1]
        dz = torch.empty((n, hs), dtype=torch.float32, device=dev)
        K4 = K + (-K) % 4
        # block rows: only every atom's own block of the gradient is defined -- all the convolution's backward reads
        gsim = torch.empty((n, K4), dtype=torch.float32, device=dev)[:, :K] if ctx.needs_input_grad[0] else None
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        gb1 = torch.empty_like(b1) if b1 is not None else None
        gb2 = torch.empty_like(b2) if b2 is not None else None
        p = _params(w1, b1, w2, b2)
        bk = _sel_buckets(plan)
        rowptr, col = plan.csr_out
        with torch.cuda.device(dev):
            ws_bytes = int(lib.mkgnn_readout_blocks_workspace_bytes(K, H, G, seg.size))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.mkgnn_readout_blocks_backward(
                p, sim.data_ptr(), _stride0(sim), _lib.Int32x4(*blocks), bk, n, rowptr.data_ptr(), col.data_ptr(),
                seg.mol_ptr.data_ptr(), seg.atom_mol.data_ptr(), seg.size, gate.data_ptr(), gsum.data_ptr(), pooled.data_ptr(),
                g.data_ptr(), _stride0(g), dz.data_ptr(), _lib.ptr(gsim), K4, gw1.data_ptr(), _lib.ptr(gb1),
                gw2.data_ptr(), _lib.ptr(gb2), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev)), "mkgnn_readout_blocks_backward")
        return gsim, gw1, gb1, gw2, gb2, None, None, None, None


def readout_blocks_supported(K: int, H: int, G: int, blocks) -> bool:
    return bool(_lib.load().mkgnn_readout_blocks_supported(int(K), int(H), int(G), _lib.Int32x4(*[int(b) for b in blocks])))


def readout_blocks(sim: torch.Tensor, plan, blocks, lin1: torch.nn.Linear, lin2: torch.nn.Linear,
                   dropout: Optional[torch.nn.Dropout], seg: "MoleculeSegments") -> torch.Tensor:
    """``readout(propagate_add(sim), ...)`` for the block rows ``sim`` of the LAST kernel convolution (``kernelsetconv(...,
    block_rows=True)``; ``blocks`` = its four kernel counts): the projection ``lin1`` runs on every atom's own block first
    and the propagate step carries H-wide rows (``_ReadoutBlocksFn``).  The caller checks ``readout_blocks_supported`` and
    that the molecule segments are sorted."""
    H = lin1.weight.shape[0]
    p_drop = dropout.p if (dropout is not None and dropout.training) else 0.0
    keep = None
    if p_drop > 0.0:
        keep = torch.empty((sim.shape[0], H), dtype=torch.float32, device=sim.device)
        if p_drop >= 1.0:
            keep.zero_()
        else:
            keep.bernoulli_(1.0 - p_drop).mul_(1.0 / (1.0 - p_drop))
    return _ReadoutBlocksFn.apply(sim, lin1.weight, lin1.bias, lin2.weight, lin2.bias, keep, seg, plan, tuple(blocks))


_TORCH_READOUT_OK = os.environ.get("MKGNN_TORCH_READOUT", "") == "1"


def readout(h: torch.Tensor, lin1: torch.nn.Linear, lin2: torch.nn.Linear, dropout: Optional[torch.nn.Dropout],
            batch: torch.Tensor, size: Optional[int] = None, segments: Optional["MoleculeSegments"] = None) -> torch.Tensor:
    """``global_add_pool(lin2(dropout(swish(lin1(h)))), batch, size)`` -> ``[size, G]``.  ``segments``: the molecule
    segments of ``batch`` when the caller already has them (a loader knows the molecule sizes; deriving them from
    ``batch`` costs a host synchronisation, which a captured step cannot have)."""
    _lib.require_gpu_tensor(h, "node_representation")
    seg = segments if segments is not None else molecule_segments(batch, size)
    H, F = lin1.weight.shape
    G = lin2.weight.shape[0]
    p_drop = dropout.p if (dropout is not None and dropout.training) else 0.0
    if not readout_supported(F, H, G) and not _TORCH_READOUT_OK:
        # no silent PyTorch-operator path for a shape the HIP kernels do not take: the caller decides
        raise _lib.MolKGNNLibraryError(
            f"readout: lin1 {F} -> {H}, lin2 -> {G} is outside the HIP readout kernels (F <= 128, H <= 64, G <= 64; inside "
            "MolKGNNNet the block-row readout takes up to 255 kernel columns); set MKGNN_TORCH_READOUT=1 to run this shape "
            "through PyTorch operators on the GPU instead")
    if not (seg.sorted and readout_supported(F, H, G) and h.shape[0] > 0 and seg.size > 0):
        # (an unsorted `batch` vector or an empty batch: the same formula through PyTorch operators)
        z = swish(lin1(h))
        if dropout is not None:
            z = dropout(z)
        z = lin2(z)
        return torch.zeros(seg.size, G, dtype=z.dtype, device=z.device).index_add_(0, batch, z)
    keep = None
    if p_drop > 0.0:
        # the multipliers torch's dropout would apply: 0 with probability p, else 1 / (1 - p)
        keep = torch.empty((h.shape[0], H), dtype=torch.float32, device=h.device)
        if p_drop >= 1.0:
            keep.zero_()
        else:
            keep.bernoulli_(1.0 - p_drop).mul_(1.0 / (1.0 - p_drop))
    return _ReadoutFn.apply(h, lin1.weight, lin1.bias, lin2.weight, lin2.bias, keep, seg)


# ------------------------------------------------------------------------------------------ batch norm --
class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn: torch.nn.BatchNorm1d, use_batch_stats: bool, n_valid=None, companion=None, split_out=False):
        lib = _lib.load()
        ctx.n_valid = n_valid
        x = _row_major(x if x.dtype == torch.float32 else x.float())
        n, C = x.shape
        dev = x.device
        out = torch.empty((n, C), dtype=torch.float32, device=dev)
        save_mean = torch.empty(C, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
        update = bn.training and bn.track_running_stats and bn.running_mean is not None
        rm = bn.running_mean if (update or not use_batch_stats) else None
        rv = bn.running_var if (update or not use_batch_stats) else None
        nbt = bn.num_batches_tracked if (update and bn.num_batches_tracked is not None) else None
        if nbt is not None and (nbt.dtype != torch.int64 or nbt.device != dev):
            raise _lib.MolKGNNLibraryError("num_batches_tracked must be an int64 tensor on the input's device")
        # the row norms of the output ride along for the kernel convolution that reads it next (<= 32 channels, 16-byte rows)
        inv = torch.empty(n, dtype=torch.float32, device=dev) \
            if (C <= 32 and C % 4 == 0 and _stride0(x) % 4 == 0 and x.data_ptr() % 16 == 0) else None
        with torch.cuda.device(dev):
            ws_bytes = int(lib.mkgnn_batchnorm_workspace_bytes(C))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            st, keep, cws, cws_bytes = None, None, None, 0
            if companion is not None and use_batch_stats and bn.training:
                st, keep = _bn_stats_struct(*companion)      # (statistics move in training mode only)
                cws_bytes = int(lib.mkgnn_batchnorm_stats_workspace_bytes(st.C))
                cws = torch.empty(cws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.mkgnn_batchnorm_forward_with_stats(
                x.data_ptr(), _stride0(x), n, C, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(rm), _lib.ptr(rv),
                float(bn.momentum if bn.momentum is not None else 0.0), float(bn.eps),
                int(use_batch_stats) | (2 if (split_out and inv is not None) else 0),       # (MKGNN_BN_SPLIT_ROWS)
                out.data_ptr(), C, save_mean.data_ptr(), save_invstd.data_ptr(), _lib.ptr(inv), _lib.ptr(nbt),
                _lib.ptr(n_valid), ws.data_ptr(), ws_bytes, None if st is None else ctypes.byref(st), _lib.ptr(cws), cws_bytes,
                _lib.stream_ptr(dev)), "mkgnn_batchnorm_forward_with_stats")
            del keep
        ctx.use_batch_stats = use_batch_stats
        ctx.save_for_backward(x, weight, save_mean, save_invstd)
        ctx.wrote_split = bool(split_out and inv is not None)
        if inv is None:
            inv = torch.empty(0, dtype=torch.float32, device=dev)
        ctx.mark_non_differentiable(inv)
        ctx.set_materialize_grads(False)
        return out, inv

    @staticmethod
    def backward(ctx, grad_out, _g_inv=None):
        lib = _lib.load()
        if grad_out is None:
            return None, None, None, None, None, None, None, None
        x, weight, save_mean, save_invstd = ctx.saved_tensors
        n, C = x.shape
        dev = x.device
        g = _row_major(grad_out if grad_out.dtype == torch.float32 else grad_out.float())
        gx = torch.empty((n, C), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        gw = torch.empty(C, dtype=torch.float32, device=dev) if (weight is not None and ctx.needs_input_grad[1]) else None
        gb = torch.empty(C, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(dev):
            ws_bytes = int(lib.mkgnn_batchnorm_workspace_bytes(C))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.mkgnn_batchnorm_backward(
                g.data_ptr(), _stride0(g), x.data_ptr(), _stride0(x), n, C, _lib.ptr(weight), save_mean.data_ptr(),
                save_invstd.data_ptr(), int(ctx.use_batch_stats), _lib.ptr(gx), C, _lib.ptr(gw), _lib.ptr(gb),
                _lib.ptr(ctx.n_valid), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev)), "mkgnn_batchnorm_backward")
        return gx, gw, gb, None, None, None, None, None


def _stats_supported(x: torch.Tensor, bn: torch.nn.BatchNorm1d) -> bool:
    return (x.is_cuda and x.dim() == 2 and 1 <= x.shape[1] <= 256 and x.shape[0] >= 1 and x.shape[1] == bn.num_features
            and bn.momentum is not None and bn.running_mean is not None and bn.running_mean.is_cuda)


def _bn_stats_struct(x: torch.Tensor, bn: torch.nn.BatchNorm1d, key: Optional[torch.Tensor], key_limit: Optional[torch.Tensor]):
    """The C struct of a statistics companion and the tensors it points at (to be kept alive across the call)."""
    x = _row_major(x if x.dtype == torch.float32 else x.float())
    if (key is None) != (key_limit is None):
        raise ValueError("key and key_limit come together")
    if key is not None:
        if not (key.is_cuda and key.dtype == torch.int64 and key.is_contiguous() and key.numel() == x.shape[0]):
            raise ValueError("key must be a contiguous int64 GPU tensor with one entry per row")
        if not (key_limit.is_cuda and key_limit.dtype == torch.int64 and key_limit.numel() == 1):
            raise ValueError("key_limit must be a one-element int64 tensor on the GPU")
    nbt = bn.num_batches_tracked
    if nbt is not None and (nbt.dtype != torch.int64 or nbt.device != x.device):
        raise _lib.MolKGNNLibraryError("num_batches_tracked must be an int64 tensor on the input's device")
    st = _lib.BnStats(x.data_ptr(), _stride0(x), x.shape[0], x.shape[1], _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var),
                      float(bn.momentum), _lib.ptr(nbt), _lib.ptr(key), _lib.ptr(key_limit))
    return st, (x, key, key_limit)


def update_running_stats(x: torch.Tensor, bn: torch.nn.BatchNorm1d, key: Optional[torch.Tensor] = None,
                         key_limit: Optional[torch.Tensor] = None) -> None:
    """What ``bn(x)`` does to the module's buffers in training mode, without computing ``bn(x)``: the reference calls
    ``edge_batch_norm(data.edge_attr)`` in every forward (``MolKGNNNet.py:116``) and the kernel convolution never reads the
    result -- ``running_mean``, ``running_var`` and ``num_batches_tracked`` still move, and they are
    state-dict contents.  ``key`` / ``key_limit``: only rows with ``key[r] < key_limit`` count (padded batches: the bonds of
    real atoms).  No-op in eval mode or without tracked statistics."""
    if not (bn.training and bn.track_running_stats and bn.running_mean is not None):
        return
    _lib.require_gpu_tensor(x, "x")
    if not _stats_supported(x, bn):
        if key is not None:
            if torch.cuda.is_current_stream_capturing():
                # (a boolean-mask index synchronises with the host: illegal inside a capture -- say so instead of failing obscurely)
                raise _lib.MolKGNNLibraryError("update_running_stats: a keyed batch outside the HIP kernels' limits (> 256 channels, "
                                               "momentum None, statistics off the GPU) cannot run inside a hipGraph capture")
            x = x[key < key_limit]
        bn(x)                                             # (PyTorch's operator on the GPU: cumulative-average momentum, > 256 channels)
        return
    lib = _lib.load()
    dev = x.device
    st, keep = _bn_stats_struct(x, bn, key, key_limit)
    with torch.cuda.device(dev):
        ws_bytes = int(lib.mkgnn_batchnorm_stats_workspace_bytes(st.C))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.mkgnn_batchnorm_update_stats(ctypes.byref(st), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev)),
                   "mkgnn_batchnorm_update_stats")
    del keep


def batch_norm(x: torch.Tensor, bn: torch.nn.BatchNorm1d, n_valid: Optional[torch.Tensor] = None,
               companion=None, split_out: bool = False) -> torch.Tensor:
    """``bn(x)`` for a 2-D input on the GPU, with ``torch.nn.BatchNorm1d``'s semantics (batch statistics in
    training mode or when no running statistics are tracked; running statistics updated in place).

    ``companion`` = ``(x2, bn2, key, key_limit)``: ``update_running_stats(x2, bn2, key, key_limit)`` done by extra blocks of
    this batch norm's own launches (training mode).

    ``split_out`` (round 6): write the result as pre-split rows (``functional.ROWS_SPLIT``) where the kernels can (<= 32 channels,
    a multiple of 4, 16-byte rows) -- for a caller whose only reader of it is a kernel convolution that takes them.

    ``n_valid`` (a one-element int64 CUDA tensor): only the leading ``n_valid`` rows enter the batch statistics; the
    remaining rows are padding (``molkgnn_amd.padding``) -- normalised with the same statistics, excluded from every sum.
    The value is read on the device, so a captured step serves batches with different numbers of real atoms."""
    _lib.require_gpu_tensor(x, "x")
    if n_valid is not None and not (n_valid.is_cuda and n_valid.dtype == torch.int64 and n_valid.numel() == 1):
        raise ValueError("n_valid must be a one-element int64 tensor on the GPU")
    if x.dim() != 2 or x.shape[1] != bn.num_features:
        raise ValueError(f"expected a [N, {bn.num_features}] input, got {tuple(x.shape)}")
    use_batch_stats = bn.training or bn.running_mean is None
    if companion is not None:
        x2, bn2 = companion[0], companion[1]
        if not (bn2.training and bn2.track_running_stats and bn2.running_mean is not None):
            companion = None                              # nothing of bn2 moves
        elif not (_stats_supported(x2, bn2) and bn.training and use_batch_stats):
            update_running_stats(*companion)              # (on its own)
            companion = None
    if x.shape[1] > 256 or x.shape[0] == 0 or (bn.training and bn.track_running_stats and bn.momentum is None):
        if n_valid is not None:
            raise ValueError("n_valid needs the HIP batch norm (<= 256 channels, momentum set)")
        if companion is not None:
            update_running_stats(*companion)
        return bn(x)
    if use_batch_stats and bn.training and x.shape[0] == 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    out, inv = _BatchNormFn.apply(x, bn.weight, bn.bias, bn, use_batch_stats, n_valid, companion, split_out)
    if inv.numel():
        from .functional import _INV_ATTR, mark_rows_split
        setattr(out, _INV_ATTR, (inv, out._version))
        if split_out:
            # (``split_out``: the caller's promise that nothing but the first kernel convolution reads the result -- its rows are
            # the fp16 hi | lo halves that convolution's matrix instructions take, functional.ROWS_SPLIT)
            mark_rows_split(out)
    return out


# ------------------------------------------------------------------------------- single-task head + loss --
_HEAD_WS: dict = {}


def _head_workspace(dev, nbytes: int) -> torch.Tensor:
    """Per-device scratch for the head kernels' per-block partials.  Calls on one device are ordered on the autograd /
    capture stream, so one buffer per device is enough.  A buffer that has been handed out is never released: a
    captured graph has its address baked in, so when a larger one is needed the old ones stay referenced here."""
    held = _HEAD_WS.setdefault(str(dev), [])
    if not held or held[-1].numel() < nbytes:
        held.append(torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev))
    return held[-1]


_HEAD_RNG: dict = {}
# MKGNN_SPLIT_HEAD=1 (diagnostics): the head's forward and backward as separate launches (two each) even in a training step
_SPLIT_HEAD = os.environ.get("MKGNN_SPLIT_HEAD") == "1"


def head_rng_state(dev) -> torch.Tensor:
    """Per-device ``{seed, offset}`` (int64) of the head kernels' dropout generator.  The seed is drawn from PyTorch's
    default CPU generator the first time a device needs it, so ``torch.manual_seed`` fixes the sequence; every forward
    with dropout advances ``offset`` on the device (also inside a replayed graph)."""
    st = _HEAD_RNG.get(str(dev))
    if st is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        st = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
        _HEAD_RNG[str(dev)] = st
    return st


def reset_head_rng(dev=None, seed=None) -> None:
    """Forget the generator state (of one device, or all): the next use draws a new seed, or uses ``seed``."""
    for k in ([str(dev)] if dev is not None else list(_HEAD_RNG)):
        _HEAD_RNG.pop(k, None)
        if seed is not None:
            _HEAD_RNG[k] = torch.tensor([int(seed), 0], dtype=torch.int64, device=torch.device(k))


_UNIT_SEEDS: list = []        # (tensor, its version counter when registered): held, so the address cannot be reused


def register_unit_gradient(one: torch.Tensor) -> None:
    """Tell the head that ``one`` is a resident tensor holding 1.0 which seeds ``backward`` (``train.backward`` does):
    when the loss's incoming gradient IS that tensor, the gradients the fused forward already wrote are the answer and the
    backward launches nothing.  Any other incoming gradient scales them.  The tensor is kept alive here (its storage
    address stays its own) and a later in-place write to it (a changed version counter) retires the registration."""
    if not any(t is one for t, _ in _UNIT_SEEDS):
        _UNIT_SEEDS.append((one, one._version))


def _is_unit_seed(grad: torch.Tensor) -> bool:
    for t, version in _UNIT_SEEDS:
        if t._version == version and t.device == grad.device and (grad is t or (
                grad.data_ptr() == t.data_ptr() and grad.numel() == 1 and grad.dtype == t.dtype)):
            return True
    return False


class _BceHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, weight, bias, target, p_drop, n_rows):
        lib = _lib.load()
        emb = _row_major(emb if emb.dtype == torch.float32 else emb.float())
        ctx.rows_total = emb.shape[0]
        B, H = int(n_rows), emb.shape[1]         # the leading n_rows rows enter the loss (the rest: padding molecules)
        dev = emb.device
        w = weight.reshape(-1).contiguous()
        y = target.reshape(-1).float().contiguous()
        pred = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        rng = head_rng_state(dev) if p_drop > 0.0 else None
        used = torch.empty(2, dtype=torch.int64, device=dev) if p_drop > 0.0 else None
        ctx.unit = None
        if any(ctx.needs_input_grad[:3]) and not _SPLIT_HEAD:
            # a training step: forward and the gradients for d loss = 1 in the same two launches (mkgnn_bce_head_fused)
            gemb = None
            if ctx.needs_input_grad[0]:
                gemb = torch.empty((ctx.rows_total, H), dtype=torch.float32, device=dev)
                if ctx.rows_total > B:               # rows beyond B (padding molecules) get a zero gradient
                    gemb[B:].zero_()
            gw = torch.empty(H, dtype=torch.float32, device=dev)
            gb = torch.empty(1, dtype=torch.float32, device=dev) if bias is not None else None
            with torch.cuda.device(dev):
                ws = _head_workspace(dev, int(lib.mkgnn_bce_head_workspace_bytes(B, H)))
                _lib.check(lib.mkgnn_bce_head_fused(
                    emb.data_ptr(), _stride0(emb), B, H, w.data_ptr(), _lib.ptr(bias), y.data_ptr(), float(p_drop),
                    _lib.ptr(rng), _lib.ptr(used), pred.data_ptr(), loss.data_ptr(), _lib.ptr(gemb), H, gw.data_ptr(),
                    _lib.ptr(gb), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)), "mkgnn_bce_head_fused")
            ctx.unit = (gemb, gw, gb)
            # (references only: a second backward over a retained graph takes the separate backward kernel below)
            ctx.save_for_backward(emb, w, y, pred, used)
            ctx.wshape = weight.shape
            ctx.has_bias = bias is not None
            ctx.p_drop = float(p_drop)
            return loss
        with torch.cuda.device(dev):
            nbytes = int(lib.mkgnn_bce_head_workspace_bytes(B, H))
            ws = _head_workspace(dev, nbytes)
            _lib.check(lib.mkgnn_bce_head_dropout_forward(
                emb.data_ptr(), _stride0(emb), B, H, w.data_ptr(), _lib.ptr(bias), y.data_ptr(), float(p_drop),
                _lib.ptr(rng), _lib.ptr(used), pred.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(dev)), "mkgnn_bce_head_dropout_forward")
        ctx.save_for_backward(emb, w, y, pred, used)
        ctx.wshape = weight.shape
        ctx.has_bias = bias is not None
        ctx.p_drop = float(p_drop)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        if ctx.unit is not None:
            gemb, gw, gb = ctx.unit
            ctx.unit = None
            if not _is_unit_seed(grad_loss):                 # d loss is not the registered 1: scale
                gl = grad_loss.reshape(()).float()
                gemb = None if gemb is None else gemb * gl
                gw = gw * gl
                gb = None if gb is None else gb * gl
            return gemb, gw.reshape(ctx.wshape), gb, None, None, None
        lib = _lib.load()
        emb, w, y, pred, used = ctx.saved_tensors
        B, H = y.numel(), emb.shape[1]
        dev = emb.device
        gl = grad_loss.reshape(1).float().contiguous()
        gemb = None
        if ctx.needs_input_grad[0]:
            # rows beyond B (padding molecules) get a zero gradient: one small fill, only when there are any
            gemb = torch.empty((ctx.rows_total, H), dtype=torch.float32, device=dev)
            if ctx.rows_total > B:
                gemb[B:].zero_()
        gw = torch.empty(H, dtype=torch.float32, device=dev)
        gb = torch.empty(1, dtype=torch.float32, device=dev) if ctx.has_bias else None
        with torch.cuda.device(dev):
            ws = _head_workspace(dev, int(lib.mkgnn_bce_head_workspace_bytes(B, H)))
            _lib.check(lib.mkgnn_bce_head_dropout_backward(
                emb.data_ptr(), _stride0(emb), B, H, w.data_ptr(), y.data_ptr(), pred.data_ptr(), gl.data_ptr(),
                ctx.p_drop, _lib.ptr(used), _lib.ptr(gemb), H, gw.data_ptr(), _lib.ptr(gb), ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(dev)), "mkgnn_bce_head_dropout_backward")
        return gemb, gw.reshape(ctx.wshape), gb, None, None, None


def bce_head_loss(emb: torch.Tensor, ffn: torch.nn.Linear, target: torch.Tensor, dropout_p: float = 0.0,
                  n_rows: Optional[int] = None) -> torch.Tensor:
    """``BCEWithLogitsLoss()(ffn(dropout(emb)).view(-1), target.view(-1).float())`` for a one-output ``ffn`` (reference
    ``model.py:147-150, 169, 190-198``) as one forward and one backward kernel; ``dropout_p`` is the probability of
    zeroing an element of ``emb`` (``nn.Dropout(ffn_dropout_rate)`` in training mode; 0 otherwise), its mask drawn
    inside the kernels from ``head_rng_state``.  ``n_rows``: only the leading rows of ``emb`` enter the loss (a padded batch's
    real molecules); the others get a zero gradient."""
    _lib.require_gpu_tensor(emb, "graph_embedding")
    n_rows = emb.shape[0] if n_rows is None else int(n_rows)
    if ffn.out_features != 1 or emb.dim() != 2 or n_rows <= 0 or n_rows > emb.shape[0] or target.numel() != n_rows:
        raise ValueError("bce_head_loss needs a one-output linear layer and one target per (leading) row")
    if not 0.0 <= dropout_p < 1.0:
        raise ValueError(f"dropout probability {dropout_p} outside [0, 1)")
    return _BceHeadFn.apply(emb, ffn.weight, ffn.bias, target, float(dropout_p), n_rows)


# ------------------------------------------------------------------- the tail of a training step, fused --
# MKGNN_FUSED_TAIL=0: readout_blocks + bce_head_loss as separate operators (nine launches; A/B, diagnostics)
_FUSED_TAIL = os.environ.get("MKGNN_FUSED_TAIL", "1") != "0"
_TAIL_WS: dict = {}


def _tail_workspace(dev, nbytes: int) -> torch.Tensor:
    """Per-device scratch of the fused tail (z and d z rows, gradient slabs); never released (a captured graph has its address
    baked in)."""
    held = _TAIL_WS.setdefault(str(dev), [])
    if not held or held[-1].numel() < nbytes:
        held.append(torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev))
    return held[-1]


def _tail_limits_ok(seg: "MoleculeSegments", plan) -> bool:
    """No molecule beyond a chunk of the fused tail (``MKGNN_TAIL_MAX_ATOMS`` atoms, ``MKGNN_TAIL_MAX_EDGES`` edges each way).
    Needs the molecule sizes on the host: one synchronisation the first time a batch is seen, remembered on its segments
    (``max_atoms`` / ``max_edges`` may also be handed over by a loader that knows them).  Inside a capture an unknown
    batch is answered with False: the separate operators take it."""
    ma, me = getattr(seg, "max_atoms", None), getattr(seg, "max_edges", None)
    if ma is None or me is None:
        if getattr(seg, "from_loader", False) or torch.cuda.is_current_stream_capturing():
            return False                                  # (refillable tensors without a loader's bound: never -- eager or captured)
        ptr = seg.mol_ptr.long()
        sizes = ptr[1:] - ptr[:-1]
        rin, rout = plan.csr_in[0].long(), plan.csr_out[0].long()
        ein, eout = rin[ptr[1:]] - rin[ptr[:-1]], rout[ptr[1:]] - rout[ptr[:-1]]
        ma, me = int(sizes.max().item()), int(torch.maximum(ein.max(), eout.max()).item())
        seg.max_atoms, seg.max_edges = ma, me
    return ma <= _lib.TAIL_MAX_ATOMS and me <= _lib.TAIL_MAX_EDGES


def tail_supported(K: int, H: int, G: int, blocks) -> bool:
    return bool(_lib.load().mkgnn_tail_supported(int(K), int(H), int(G), _lib.Int32x4(*[int(b) for b in blocks])))


# Round 6: inside ``deferred_tail_reduce()`` the fused tail leaves its last launch -- the fixed-order reduction that writes the loss
# and the six parameter gradients, which nothing before the optimiser reads -- to the kernel convolution's backward, which makes it
# on its helper stream in front of the bank kernel (``mkgnn_tail_args.defer_reduce``): 10 us off the chain the next layer's
# gradient waits for.  Only ``train.training_step`` / ``train.CapturedSteps`` / bench.py open such a region, around
# ``loss -> backward`` as ONE unit, and flush at its end; ``model.loss`` anywhere else returns a finished loss as before.
_DEFER_TAIL_REDUCE = False
_TAIL_HELD: list = []        # every tensor a pending reduction still reads or writes: referenced until the flush (the reduction runs on
                             # another stream, later than the allocator's stream-ordered reuse of a freed block would allow)


def tail_flush(device) -> None:
    """``mkgnn_tail_flush``: a reduction the fused tail left pending on this thread is launched on the current stream now."""
    try:
        with torch.cuda.device(device):
            _lib.check(_lib.load().mkgnn_tail_flush(_lib.stream_ptr(device)), "mkgnn_tail_flush")
    finally:
        _TAIL_HELD.clear()           # (launched, here or by the backward's helper stream which the caller's stream has joined)


class deferred_tail_reduce:
    """``with deferred_tail_reduce(device): loss = model.loss(batch); backward(loss)`` -- see ``_DEFER_TAIL_REDUCE``.  Leaving the
    block launches whatever is still pending, on the current stream: behind it the loss and every gradient are complete in
    stream order, exactly as without the block.  Inside it the loss tensor must not be read."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        global _DEFER_TAIL_REDUCE
        self.prev, _DEFER_TAIL_REDUCE = _DEFER_TAIL_REDUCE, _TAIL_DEFER_ENABLED
        return self

    def __exit__(self, *exc):
        global _DEFER_TAIL_REDUCE
        _DEFER_TAIL_REDUCE = self.prev
        if not self.prev and self.device is not None and torch.device(self.device).type == "cuda":
            tail_flush(self.device)
        return False


_TAIL_DEFER_ENABLED = os.environ.get("MKGNN_TAIL_DEFER", "1") != "0"


class _TailFn(torch.autograd.Function):
    """``BCEWithLogitsLoss()(ffn(dropout(readout(propagate(sim)))), y)`` with every gradient for d loss = 1 in the same launch
    (``mkgnn_tail_fused``); the backward hands them out (scaled, if the incoming gradient is not the registered unit seed)."""

    @staticmethod
    def forward(ctx, sim, w1, b1, w2, b2, wh, bh, target, seg, plan, blocks, p_drop, n_rows):
        lib = _lib.load()
        _lib.require_gpu_tensor(sim, "sim_sc")
        n, K = sim.shape
        dev = sim.device
        w1c, w2c = w1.contiguous(), w2.contiguous()
        H, G = w1c.shape[0], w2c.shape[0]
        whc = wh.reshape(-1).contiguous()
        y = target.reshape(-1).float().contiguous()
        B = int(n_rows)
        pred = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        K4 = K + (-K) % 4
        gsim = torch.empty((n, K4), dtype=torch.float32, device=dev)[:, :K]     # block rows: only every atom's own block is written
        gw1, gw2, gwh = torch.empty_like(w1c), torch.empty_like(w2c), torch.empty_like(whc)
        gb1 = torch.empty_like(b1) if b1 is not None else None
        gb2 = torch.empty_like(b2) if b2 is not None else None
        gbh = torch.empty(1, dtype=torch.float32, device=dev) if bh is not None else None
        rng = head_rng_state(dev) if p_drop > 0.0 else None
        used = torch.empty(2, dtype=torch.int64, device=dev) if p_drop > 0.0 else None
        a = _lib.TailArgs()
        a.sim, a.sim_stride = sim.data_ptr(), _stride0(sim)
        for i, L in enumerate(blocks):
            a.num_kernels[i] = int(L)
        bk = _sel_buckets(plan)
        a.buckets = ctypes.cast(bk, ctypes.c_void_p)
        (rin, cin), (rout, cout) = plan.csr_in, plan.csr_out
        a.in_rowptr, a.in_col, a.out_rowptr, a.out_col = rin.data_ptr(), cin.data_ptr(), rout.data_ptr(), cout.data_ptr()
        a.mol_ptr, a.atom_mol = seg.mol_ptr.data_ptr(), seg.atom_mol.data_ptr()
        a.n_atoms, a.n_mols, a.n_loss_mols = n, seg.size, B
        a.readout = _params(w1c, b1, w2c, b2)
        a.head_weight, a.head_bias, a.target = whc.data_ptr(), _lib.ptr(bh), y.data_ptr()
        a.dropout_p, a.rng_state, a.rng_used = float(p_drop), _lib.ptr(rng), _lib.ptr(used)
        a.emb, a.emb_stride = None, 0
        a.pred, a.loss = pred.data_ptr(), loss.data_ptr()
        a.grad_sim, a.grad_sim_stride = gsim.data_ptr(), K4
        a.grad_lin1_weight, a.grad_lin1_bias = gw1.data_ptr(), _lib.ptr(gb1)
        a.grad_lin2_weight, a.grad_lin2_bias = gw2.data_ptr(), _lib.ptr(gb2)
        a.grad_head_weight, a.grad_head_bias = gwh.data_ptr(), _lib.ptr(gbh)
        # (deferred only where a backward will follow in the same region: the caller of deferred_tail_reduce promises it)
        ctx.deferred = bool(_DEFER_TAIL_REDUCE and ctx.needs_input_grad[0])
        a.defer_reduce = 1 if ctx.deferred else 0
        ctx.params = (w1, b1, w2, b2, wh, bh)
        ctx.dev = dev
        with torch.cuda.device(dev):
            ws = _tail_workspace(dev, int(lib.mkgnn_tail_workspace_bytes(K, H, G, n, seg.size)))
            _lib.check(lib.mkgnn_tail_fused(ctypes.byref(a), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)), "mkgnn_tail_fused")
        if ctx.deferred:
            # (their STORAGES: a second reference to a gradient TENSOR would make autograd copy it into .grad instead of adopting
            # it -- a copy made before the reduction has written it)
            _TAIL_HELD.extend(t.untyped_storage() for t in (loss, gw1, gb1, gw2, gb2, gwh, gbh, rng, used, ws) if t is not None)
        ctx.unit = (gsim, gw1, gb1, gw2, gb2, gwh.reshape(wh.shape), gbh)
        ctx.pred = pred
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        if ctx.unit is None:
            raise RuntimeError("the fused tail keeps its gradients for ONE backward (retain_graph is not supported: "
                               "MKGNN_FUSED_TAIL=0 runs the separate operators)")
        grads, ctx.unit = ctx.unit, None
        params, ctx.params = ctx.params, None
        # a deferred reduction has not written the parameter gradients yet: fine as long as nothing on this stream touches them
        # before the region's flush -- a scale by the incoming gradient or an accumulation into an existing .grad would
        if ctx.deferred and (not _is_unit_seed(grad_loss) or torch.is_grad_enabled()
                             or any(p is not None and p.grad is not None for p in params)):
            tail_flush(ctx.dev)
        if not _is_unit_seed(grad_loss):                     # d loss is not the registered 1: scale
            gl = grad_loss.reshape(()).float()
            grads = tuple(None if g is None else g * gl for g in grads)
        return (*grads, None, None, None, None, None, None)


def tail_loss(sim: torch.Tensor, plan, blocks, lin1: torch.nn.Linear, lin2: torch.nn.Linear, ffn: torch.nn.Linear,
              target: torch.Tensor, seg: "MoleculeSegments", dropout_p: float = 0.0, n_rows: Optional[int] = None) -> torch.Tensor:
    """``bce_head_loss(readout_blocks(sim, ...), ffn, target, dropout_p, n_rows)`` as ONE operator whose forward also takes
    every gradient (``_TailFn``).  The caller has checked ``tail_supported``, ``_tail_limits_ok`` and that the readout has
    no dropout of its own."""
    n_rows = seg.size if n_rows is None else int(n_rows)
    if ffn.out_features != 1 or n_rows <= 0 or n_rows > seg.size or target.numel() != n_rows:
        raise ValueError("tail_loss needs a one-output linear layer and one target per (leading) molecule")
    return _TailFn.apply(sim, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ffn.weight, ffn.bias, target, seg, plan,
                         tuple(blocks), float(dropout_p), n_rows)
