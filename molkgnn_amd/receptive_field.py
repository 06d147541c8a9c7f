"""Degree-bucketed receptive fields: the input layout of the kernel convolution.

This is the host-side builder for the per-degree tensors the hot path consumes.
It restates, as one vectorised pass over a whole (already collated) batch, what
the reference computes per molecule in ``ToXAndPAndEdgeAttrForDeg``
(reference ``wrapper.py:559-672``) followed by PyG collation:

* ``selected_index_degD [N_d]``  atoms whose out-degree is ``D`` in ascending
  node id (``wrapper.py:574-576, 600-601``);
* ``nei_index_degD [N_d*D]``     for every such atom the targets of the edges
  whose source it is, **in edge-list order** (``wrapper.py:567-572, 623-624``);
* ``nei_p_degD [N_d, D, 3]``, ``p_focal_degD [N_d, 3]``;
* ``nei_edge_attr_degD [N_d, D, E]`` the raw attributes of bond ``2*(e//2)`` for
  each such edge ``e`` (``wrapper.py:578-593``);
* an absent degree gives empty tensors (``wrapper.py:627-630``).

Building on the batch equals per-molecule build + collate because the edge
list is block diagonal and PyG offsets every ``*index*`` key by the molecule's
node offset.

``build_receptive_fields`` is torch index arithmetic (sort, bincount, nonzero)
and runs wherever the batch lives; it is the definition the tests check against
a per-atom brute force.  ``build_receptive_fields_hip`` builds the same tensors
with two HIP passes behind the C ABI (``mkgnn_rf_count`` / ``mkgnn_rf_fill``,
SURVEY.md 8 f-2) and is what ``attach_receptive_fields`` uses for GPU batches.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

MAX_DEGREE = 4


class GraphBatch:
    """Attribute bag with the reference's ``Data`` field names.

    The hot path only ever reads attributes from the ``data`` object it is
    handed (reference ``kernels.py:622-646``), so a plain attribute bag is a
    drop-in for ``torch_geometric.data.Data`` there.
    """

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def to(self, device, non_blocking: bool = False):
        out = GraphBatch()
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                v = v.to(device, non_blocking=non_blocking)
            setattr(out, k, v)
        return out


def build_receptive_fields(x: torch.Tensor, p: torch.Tensor,
                           edge_index: torch.Tensor,
                           edge_attr: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Return the 20 per-degree tensors (names as in the reference) for a batch.

    ``edge_index`` is ``[2, M]`` int64 with each bond stored as two consecutive
    directed edges carrying identical attributes (reference ``wrapper.py:152-156``).
    """
    n = x.shape[0]
    src = edge_index[0]
    dst = edge_index[1]
    m = src.shape[0]
    dev = src.device
    deg = torch.bincount(src, minlength=n)
    # edges grouped by source, original edge order preserved inside a group
    order = torch.sort(src, stable=True).indices
    rowptr = torch.zeros(n + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    out: Dict[str, torch.Tensor] = {}
    for d in range(1, MAX_DEGREE + 1):
        sel = (deg == d).nonzero(as_tuple=True)[0]
        if sel.numel() == 0:
            out[f"p_focal_deg{d}"] = p.new_zeros((0, p.shape[1]))
            out[f"nei_p_deg{d}"] = p.new_zeros((0,))
            out[f"nei_edge_attr_deg{d}"] = edge_attr.new_zeros((0,))
            out[f"selected_index_deg{d}"] = sel
            out[f"nei_index_deg{d}"] = torch.zeros((0,), dtype=torch.long, device=dev)
            continue
        pos = rowptr[sel].unsqueeze(1) + torch.arange(d, device=dev).unsqueeze(0)
        eid = order[pos]                      # [N_d, d] original edge ids
        nei = dst[eid]                        # [N_d, d]
        bond = 2 * torch.div(eid, 2, rounding_mode="floor")
        out[f"p_focal_deg{d}"] = p[sel]
        out[f"nei_p_deg{d}"] = p[nei]
        out[f"nei_edge_attr_deg{d}"] = edge_attr[bond]
        out[f"selected_index_deg{d}"] = sel
        out[f"nei_index_deg{d}"] = nei.reshape(-1)
    return out


def check_sizes(rf) -> None:
    """Raise if a builder run with ``sizes=`` met a batch whose real degree counts differ (one host round trip; call it
    outside captures, e.g. once per epoch or in tests).  ``rf``: the builder's dict, or the batch it was attached to."""
    get = rf.get if isinstance(rf, dict) else (lambda k: getattr(rf, k, None))
    counts, sizes = get("rf_counts"), get("rf_sizes")
    if counts is None or sizes is None:
        return
    real = [int(v) for v in counts.tolist()]
    if len(real) > 4:                            # mkgnn_index_build: [N_1..N_4, atoms that did not fit a bucket, -]
        if real[4] != 0:
            raise ValueError(f"{real[4]} atoms did not fit the bucket sizes {list(sizes)} the index was built for (their rows were dropped)")
        real = real[:4]
    if real != [int(v) for v in sizes]:
        raise ValueError(f"receptive fields were built for bucket sizes {list(sizes)} but the batch has {real} atoms of "
                         "degree 1..4 (rows beyond the real sizes are zero-filled, rows beyond the given sizes were dropped)")


def build_receptive_fields_hip(x: torch.Tensor, p: torch.Tensor, edge_index: torch.Tensor,
                               edge_attr: torch.Tensor, sizes=None, validate: bool = False) -> Dict[str, torch.Tensor]:
    """Same result as ``build_receptive_fields`` (bit for bit) for a batch on the GPU, through the C ABI.

    ``sizes`` = the four bucket sizes when the caller knows them (a batch padded to a fixed shape): no host round trip,
    so the build can run inside a captured graph; the degree counts of the batch must be exactly these.  If they are
    not, atoms beyond a given size are dropped and rows beyond a real size are zero-filled (never uninitialised); the
    result then carries ``rf_counts`` (device, the real sizes) / ``rf_sizes`` for ``check_sizes``, which ``validate=True``
    calls at once (a host synchronisation)."""
    from . import _lib
    lib = _lib.load()
    _lib.require_gpu_tensor(edge_index, "edge_index")
    dev = edge_index.device
    n, m = x.shape[0], edge_index.shape[1]
    ei = edge_index.contiguous().long()
    pf = p.contiguous().float()
    ea = edge_attr.contiguous().float()
    E = ea.shape[1] if ea.dim() == 2 else 1
    with torch.cuda.device(dev):
        st = _lib.stream_ptr(dev)
        ws = torch.empty(int(lib.mkgnn_rf_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        counts = torch.empty(4, dtype=torch.int64, device=dev)
        _lib.check(lib.mkgnn_rf_count(_lib.ptr(ei), n, m, ws.data_ptr(), ws.numel(), counts.data_ptr(), st), "mkgnn_rf_count")
        given = sizes is not None
        if sizes is None:
            sizes = counts.tolist()              # the one host round trip: the outputs have to be allocated
        out: Dict[str, torch.Tensor] = {}
        if given:
            out["rf_counts"], out["rf_sizes"] = counts, tuple(int(v) for v in sizes)
        buckets = _lib.Buckets4()
        raw = {}
        for d in range(1, MAX_DEGREE + 1):
            nd = int(sizes[d - 1])
            if nd == 0:
                continue
            sel = torch.empty(nd, dtype=torch.long, device=dev)
            nei = torch.empty(nd * d, dtype=torch.long, device=dev)
            nea = torch.empty((nd, d, E), dtype=torch.float32, device=dev)
            pfo = torch.empty((nd, 3), dtype=torch.float32, device=dev)
            pne = torch.empty((nd, d, 3), dtype=torch.float32, device=dev)
            # the unit-normalised bond rows ride along (the kernel convolution keeps them per batch: plan.Bucket.e_unit)
            neu = torch.empty((nd * d, 8), dtype=torch.float32, device=dev) if (ea.dim() == 2 and E <= 8) else None
            b = buckets[d - 1]
            b.count = nd
            b.selected_index, b.nei_index = sel.data_ptr(), nei.data_ptr()
            b.nei_edge_attr, b.p_focal, b.nei_p = nea.data_ptr(), pfo.data_ptr(), pne.data_ptr()
            b.nei_edge_unit = _lib.ptr(neu)
            raw[d] = (sel, nei, nea, pfo, pne, neu)
        _lib.check(lib.mkgnn_rf_fill(_lib.ptr(ei), pf.data_ptr(), _lib.ptr(ea), n, m, E, ws.data_ptr(), buckets, st),
                   "mkgnn_rf_fill")
        for d in range(1, MAX_DEGREE + 1):
            if d not in raw:                     # shapes of the reference's empty fields (wrapper.py:627-630)
                out[f"p_focal_deg{d}"] = p.new_zeros((0, p.shape[1]))
                out[f"nei_p_deg{d}"] = p.new_zeros((0,))
                out[f"nei_edge_attr_deg{d}"] = edge_attr.new_zeros((0,))
                out[f"selected_index_deg{d}"] = torch.zeros((0,), dtype=torch.long, device=dev)
                out[f"nei_index_deg{d}"] = torch.zeros((0,), dtype=torch.long, device=dev)
                continue
            sel, nei, nea, pfo, pne, neu = raw[d]
            if neu is not None and edge_attr.dtype == torch.float32:
                out[f"nei_edge_unit_deg{d}"] = neu
            out[f"p_focal_deg{d}"] = pfo.to(p.dtype)
            out[f"nei_p_deg{d}"] = pne.to(p.dtype)
            out[f"nei_edge_attr_deg{d}"] = nea.to(edge_attr.dtype) if ea.dim() == 2 else nea.to(edge_attr.dtype).reshape(sel.numel(), d)
            out[f"selected_index_deg{d}"] = sel
            out[f"nei_index_deg{d}"] = nei
    if given and validate:
        check_sizes(out)
    return out


def build_index_hip(x: torch.Tensor, p: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor, sizes, rf_event=None):
    """Receptive fields AND index plan of a batch whose bucket sizes are known (``mkgnn_index_build``: one memset and six
    kernels for what ``build_receptive_fields_hip(sizes=...)`` + ``BatchPlan.build_hip`` do in two and nine; round 5).
    Returns ``(rf, plan_parts)``: ``rf`` as ``build_receptive_fields_hip`` gives it (``rf_counts`` has six entries here:
    the real sizes, then the number of atoms that did not fit a bucket), ``plan_parts`` the tensors of a ``BatchPlan``
    (``plan.plan_from_lists_cached(..., prebuilt=plan_parts)``).  Entry for entry the separate builders' results.
    ``rf_event`` (a ``torch.cuda.Event`` that has been recorded once, so that it exists): re-recorded inside the call where
    the receptive fields are complete -- two plan kernels earlier than the end of the call."""
    from . import _lib
    lib = _lib.load()
    _lib.require_gpu_tensor(edge_index, "edge_index")
    dev = edge_index.device
    n, m = x.shape[0], edge_index.shape[1]
    ei = edge_index.contiguous().long()
    pf = p.contiguous().float()
    ea = edge_attr.contiguous().float()
    if ea.dim() != 2 or pf.shape[1] != 3 or n < 1:
        raise ValueError("build_index_hip: needs [M, E] bond attributes, [N, 3] coordinates and at least one atom")
    E = ea.shape[1]
    sizes = tuple(int(v) for v in sizes)
    r = sum(nd * (d + 1) for d, nd in zip(range(1, 5), sizes))
    out: Dict[str, torch.Tensor] = {}
    with torch.cuda.device(dev):
        counts = torch.empty(6, dtype=torch.int64, device=dev)
        out["rf_counts"], out["rf_sizes"] = counts, sizes
        buckets = _lib.Buckets4()
        raw = {}
        for d in range(1, MAX_DEGREE + 1):
            nd = sizes[d - 1]
            if nd == 0:
                continue
            sel = torch.empty(nd, dtype=torch.long, device=dev)
            nei = torch.empty(nd * d, dtype=torch.long, device=dev)
            nea = torch.empty((nd, d, E), dtype=torch.float32, device=dev)
            pfo = torch.empty((nd, 3), dtype=torch.float32, device=dev)
            pne = torch.empty((nd, d, 3), dtype=torch.float32, device=dev)
            neu = torch.empty((nd * d, 8), dtype=torch.float32, device=dev) if E <= 8 else None
            b = buckets[d - 1]
            b.count = nd
            b.selected_index, b.nei_index = sel.data_ptr(), nei.data_ptr()
            b.nei_edge_attr, b.p_focal, b.nei_p = nea.data_ptr(), pfo.data_ptr(), pne.data_ptr()
            b.nei_edge_unit = _lib.ptr(neu)
            raw[d] = (sel, nei, nea, pfo, pne, neu)
        i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=dev)      # noqa: E731
        s_ptr, s_rows = i32(n + 1), i32(r)
        in_ptr, in_col, in_pk, out_ptr, out_col = i32(n + 1), i32(m), i32(m), i32(n + 1), i32(m)
        deg8 = torch.empty(n, dtype=torch.int8, device=dev)
        nbytes = int(lib.mkgnn_index_workspace_bytes(n, m, r))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.mkgnn_index_build(_lib.ptr(ei), pf.data_ptr(), _lib.ptr(ea), n, m, E, buckets, s_ptr.data_ptr(), s_rows.data_ptr(),
                                         in_ptr.data_ptr(), in_col.data_ptr(), in_pk.data_ptr(), out_ptr.data_ptr(), out_col.data_ptr(),
                                         deg8.data_ptr(), counts.data_ptr(), ws.data_ptr(), nbytes,
                                         None if rf_event is None else rf_event.cuda_event, _lib.stream_ptr(dev)),
                   "mkgnn_index_build")
    for d in range(1, MAX_DEGREE + 1):
        if d not in raw:                         # shapes of the reference's empty fields (wrapper.py:627-630)
            out[f"p_focal_deg{d}"] = p.new_zeros((0, p.shape[1]))
            out[f"nei_p_deg{d}"] = p.new_zeros((0,))
            out[f"nei_edge_attr_deg{d}"] = edge_attr.new_zeros((0,))
            out[f"selected_index_deg{d}"] = torch.zeros((0,), dtype=torch.long, device=dev)
            out[f"nei_index_deg{d}"] = torch.zeros((0,), dtype=torch.long, device=dev)
            continue
        sel, nei, nea, pfo, pne, neu = raw[d]
        if neu is not None and edge_attr.dtype == torch.float32:
            out[f"nei_edge_unit_deg{d}"] = neu
        out[f"p_focal_deg{d}"] = pfo.to(p.dtype)
        out[f"nei_p_deg{d}"] = pne.to(p.dtype)
        out[f"nei_edge_attr_deg{d}"] = nea.to(edge_attr.dtype)
        out[f"selected_index_deg{d}"] = sel
        out[f"nei_index_deg{d}"] = nei
    parts = {"scatter": (s_ptr, s_rows[:r]), "deg8": deg8, "csr_in": (in_ptr, in_col[:m]), "csr_out": (out_ptr, out_col[:m]),
             "csr_in_packed": (in_ptr, in_pk[:m]), "keep": (ws, ei, pf, ea, buckets)}
    return out, parts


def _merged_index() -> bool:
    """MKGNN_MERGED_INDEX=1: fixed-shape batches build receptive fields and plan with the one-pass builder (mkgnn_index_build).
    Not the default: measured on MI355X (round 5, batch 4096) its six kernels take what the separate builders' nine take
    (78 against 85 us of kernel time; an epoch of fresh batches 0.944 against 0.946 ms per step, the shard-fed epoch 1.06
    against 1.02) -- the chain is bound by the scattered row writes of its fill kernel, not by launches."""
    import os
    return os.environ.get("MKGNN_MERGED_INDEX", "0") == "1"


def attach_receptive_fields(batch: GraphBatch, sizes=None, overlap: bool = False) -> GraphBatch:
    """The reference's per-degree tensors of a collated batch, set on the batch object.

    ``overlap=True`` (CUDA, ``sizes`` given -- the fixed-shape path of molkgnn_amd.padding): the builder runs on the
    device's index stream (plan.index_stream) and the batch carries the event ``_rf_ready``; ``MolKGNNNet.forward`` makes
    its stream wait for it after the atom batch norm, which needs none of these tensors -- inside a captured step the
    builder runs beside the batch norm instead of in front of it."""
    if batch.edge_index.is_cuda and batch.p.shape[1] == 3:
        side = None
        if overlap and sizes is not None:
            from .plan import index_stream
            side = index_stream(batch.edge_index.device)
        if side is not None:
            cur = torch.cuda.current_stream(batch.edge_index.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                from .plan import plan_from_lists_cached
                names = ('p_focal', 'nei_p', 'nei_edge_attr', 'selected_index', 'nei_index')
                merged = _merged_index() and batch.edge_attr.dim() == 2 and batch.x.shape[0] > 0
                if merged:
                    # receptive fields and index plan from ONE pass over the batch (mkgnn_index_build); the event the first
                    # convolution waits for is recorded inside the call, where the buckets are complete
                    ev = torch.cuda.Event()
                    ev.record(side)                      # (creates the event; the call records it again further down the stream)
                    rf, parts = build_index_hip(batch.x, batch.p, batch.edge_index, batch.edge_attr, sizes, rf_event=ev)
                else:
                    rf, parts = build_receptive_fields_hip(batch.x, batch.p, batch.edge_index, batch.edge_attr, sizes), None
                    ev = torch.cuda.Event()
                    ev.record(side)
                # ... and the index plan right behind it on the same stream (it needs the buckets and edge_index, nothing of
                # the caller's stream): MolGCN.forward finds it in the plan cache; the first propagate joins
                units = [rf.get(f'nei_edge_unit_deg{d}') for d in range(1, 5)]
                plan_from_lists_cached(batch.x.shape[0], *[[rf[f'{nm}_deg{d}'] for d in range(1, 5)] for nm in names],
                                       batch.edge_index, units if any(u is not None for u in units) else None, prebuilt=parts)
            for v in rf.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
            if parts is not None:                        # (allocated on the index stream, read on the caller's)
                for v in parts.values():
                    for t in (v if isinstance(v, tuple) else (v,)):
                        if torch.is_tensor(t):
                            t.record_stream(cur)
            if not torch.cuda.is_current_stream_capturing():
                # (the builder reads the batch on the index stream: a batch dropped before anything joined must not have its
                # memory handed out again under it)
                for v in (batch.x, batch.p, batch.edge_index, batch.edge_attr):
                    v.record_stream(side)
            batch._rf_ready = ev
        elif sizes is not None and _merged_index() and batch.edge_attr.dim() == 2 and batch.x.shape[0] > 0:
            from .plan import plan_from_lists_cached
            rf, parts = build_index_hip(batch.x, batch.p, batch.edge_index, batch.edge_attr, sizes)
            names = ('p_focal', 'nei_p', 'nei_edge_attr', 'selected_index', 'nei_index')
            units = [rf.get(f'nei_edge_unit_deg{d}') for d in range(1, 5)]
            plan_from_lists_cached(batch.x.shape[0], *[[rf[f'{nm}_deg{d}'] for d in range(1, 5)] for nm in names],
                                   batch.edge_index, units if any(u is not None for u in units) else None, prebuilt=parts)
        else:
            rf = build_receptive_fields_hip(batch.x, batch.p, batch.edge_index, batch.edge_attr, sizes)
    else:
        rf = build_receptive_fields(batch.x, batch.p, batch.edge_index, batch.edge_attr)
    for k, v in rf.items():
        setattr(batch, k, v)
    return batch


def await_receptive_fields(batch) -> None:
    """Make the current stream wait for a builder that ``attach_receptive_fields(..., overlap=True)`` left running."""
    ev = getattr(batch, "_rf_ready", None)
    if ev is not None:
        batch._rf_ready = None
        torch.cuda.current_stream().wait_event(ev)
