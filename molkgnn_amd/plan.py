"""Per-batch index plan: everything the HIP kernels need that depends only on
the graph structure, computed once per batch and shared by all layers.

* the four degree buckets exactly as the reference's ``data`` object holds them
  (``selected_index_degD`` ... see receptive_field.py);
* the scatter CSR of the backward pass: for every atom, the list of
  "contribution rows" (bucket by bucket, atom by atom, focal then neighbours)
  that carry a gradient for it -- the deterministic replacement of the
  ``index_select`` backward's scatter-add (reference kernels.py:527, 543);
* the two CSR forms of ``edge_index`` for ``MolGCN.propagate``
  (KernelLayer.py:119-123): edges grouped by target (forward) and by source
  (gradient).
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional

import torch

MAX_DEGREE = 4


def _as_f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class Bucket:
    __slots__ = ("degree", "count", "sel", "nei", "e_nei", "p_focal", "nei_p", "_e_unit")

    def __init__(self, degree, sel, nei, e_nei, p_focal, nei_p, e_unit=None):
        self.degree = degree
        self.count = int(sel.numel())
        self.sel = sel.contiguous().long()
        self.nei = nei.contiguous().long()
        if self.nei.numel() != self.count * degree:
            raise ValueError(f"nei_index_deg{degree} has {self.nei.numel()} entries for {self.count} focal atoms")
        self.e_nei = _as_f32(e_nei)
        self.p_focal = _as_f32(p_focal)
        self.nei_p = _as_f32(nei_p)
        # unit bond rows that came with the receptive fields (mkgnn_rf_fill writes them next to the raw ones)
        self._e_unit = e_unit if (e_unit is not None and e_unit.numel() == self.count * degree * 8) else None

    def e_unit(self, E: int):
        """``[N_d * d, 8]`` unit-normalised bond attributes (``mkgnn_unit_rows8``), built once per batch: the bonds of a
        batch are the same in every layer and every step.  ``None`` when not applicable (CPU tensors, E > 8)."""
        if self._e_unit is None and self.count and self.e_nei is not None and self.e_nei.is_cuda and 1 <= E <= 8 \
                and self.e_nei.numel() == self.count * self.degree * E:
            from . import _lib
            rows = self.count * self.degree
            out = torch.empty((rows, 8), dtype=torch.float32, device=self.e_nei.device)
            with torch.cuda.device(out.device):
                _lib.check(_lib.load().mkgnn_unit_rows8(self.e_nei.data_ptr(), rows, E, out.data_ptr(),
                                                        _lib.stream_ptr(out.device)), "mkgnn_unit_rows8")
            self._e_unit = out
        return self._e_unit


_INDEX_STREAMS: dict = {}


def index_stream(dev):
    """The side stream (one per device) on which the index structures of a batch are built (receptive fields, plan);
    ``None`` with ``MKGNN_INDEX_OVERLAP=0``."""
    if os.environ.get("MKGNN_INDEX_OVERLAP") == "0":
        return None
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    st = _INDEX_STREAMS.get(key)
    if st is None:
        st = _INDEX_STREAMS[key] = torch.cuda.Stream(device=key)
    return st


class BatchPlan:
    def __init__(self, n_atoms: int, buckets: List[Bucket], edge_index: Optional[torch.Tensor] = None):
        self.n_atoms = int(n_atoms)
        self.buckets = buckets
        self.device = buckets[0].sel.device
        self.n_focal = sum(b.count for b in buckets)
        self.n_slots = sum(b.count * b.degree for b in buckets)
        self.edge_index = edge_index
        self._scatter = None
        self._csr_in = None
        self._csr_out = None
        self._deg8 = None
        self._csr_in_packed = None
        self._ready = None          # event of a build still running on the index stream (build_hip)
        self._build_ws = None

    def _await(self):
        """Make the current stream wait for a build that runs on the index stream (once)."""
        ev, self._ready = self._ready, None
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    # -- all index structures in one pass on the GPU (mkgnn_plan_build) ----------
    def build_hip(self) -> bool:
        """Scatter CSR, both propagate CSRs, ``deg8`` and the packed columns from one call of ``mkgnn_plan_build``
        (no sort over the batch, no host synchronisation); entry for entry what the torch definitions below give
        (``tests/test_hip_parity.py::test_plan_builder_hip_matches_torch_builder``).  False when not applicable (CPU).

        The kernels run on the device's index stream (``index_stream``), forked from the current stream; the first read of
        any of the structures joins it.  Nothing before the first ``propagate`` of a step needs them, so inside a captured
        step the build runs beside the first convolution instead of in front of it (``MKGNN_INDEX_OVERLAP=0``: in line)."""
        if not self.device.type == "cuda" or os.environ.get("MKGNN_TORCH_PLAN"):
            return False
        if self._scatter is not None:
            return True
        from . import _lib
        lib = _lib.load()
        dev, n = self.device, self.n_atoms
        ei = self.edge_index
        m = int(ei.shape[1]) if ei is not None else 0
        r = sum(b.count * (b.degree + 1) for b in self.buckets)
        if ei is not None:
            ei = ei.contiguous()
            if ei.dtype != torch.int64:
                ei = ei.long()
        bk = _lib.Buckets4()
        for i, b in enumerate(self.buckets):
            bk[i].count = b.count
            if b.count:
                bk[i].selected_index, bk[i].nei_index = b.sel.data_ptr(), b.nei.data_ptr()
        i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=dev)      # noqa: E731
        s_ptr, s_rows = i32(n + 1), i32(r)
        in_ptr, in_col, in_pk, out_ptr, out_col = i32(n + 1), i32(m), i32(m), i32(n + 1), i32(m)
        deg8 = torch.empty(max(n, 1), dtype=torch.int8, device=dev)
        with torch.cuda.device(dev):
            nbytes = int(lib.mkgnn_plan_workspace_bytes(n, m, r))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            side = index_stream(dev)
            cur = torch.cuda.current_stream(dev)
            if side is not None and side != cur:       # (already on the index stream: attach_receptive_fields(overlap=True))
                side.wait_stream(cur)
            with torch.cuda.stream(side if side is not None else cur):
                _lib.check(lib.mkgnn_plan_build(bk, n, _lib.ptr(ei), m, s_ptr.data_ptr(), s_rows.data_ptr(), in_ptr.data_ptr(),
                                                in_col.data_ptr(), in_pk.data_ptr(), out_ptr.data_ptr(), out_col.data_ptr(),
                                                deg8.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr(dev)), "mkgnn_plan_build")
                if side is not None:
                    self._ready = torch.cuda.Event()
                    self._ready.record(side)
                    self._build_ws = (ws, ei, bk)      # alive until the plan goes (its kernels run on another stream)
                    if side != cur and not torch.cuda.is_current_stream_capturing():
                        # ... and beyond: a plan that is dropped before anything joined the index stream (a loader probe, an
                        # exception) hands these blocks back to the allocator of the CALLER's stream, whose next allocation --
                        # the next batch's builder -- would overwrite the workspace under the kernels still running here
                        # (seen as a GPU memory fault once in a few hundred batches).  The allocator now waits for this stream.
                        for t in (ws, ei, s_ptr, s_rows, in_ptr, in_col, in_pk, out_ptr, out_col, deg8,
                                  *[x for b in self.buckets if b.count for x in (b.sel, b.nei)]):
                            if t is not None and t.is_cuda:
                                t.record_stream(side)
        self._scatter = (s_ptr, s_rows[:r])
        self._deg8 = deg8[:n]
        if self.edge_index is not None:
            self._csr_in, self._csr_out = (in_ptr, in_col[:m]), (out_ptr, out_col[:m])
            self._csr_in_packed = (in_ptr, in_pk[:m])
        return True

    def adopt(self, parts) -> bool:
        """Take over index structures built elsewhere (``receptive_field.build_index_hip``: the one-pass builder), on the
        stream that is current now; readers on other streams join through ``_ready`` like after ``build_hip``."""
        if self.device.type != "cuda" or self._scatter is not None:
            return self._scatter is not None
        self._scatter, self._deg8 = parts["scatter"], parts["deg8"]
        if self.edge_index is not None:
            self._csr_in, self._csr_out, self._csr_in_packed = parts["csr_in"], parts["csr_out"], parts["csr_in_packed"]
        self._build_ws = parts.get("keep")
        cur = torch.cuda.current_stream(self.device)
        side = index_stream(self.device)
        if side is not None and cur == side:
            self._ready = torch.cuda.Event()
            self._ready.record(side)
            if not torch.cuda.is_current_stream_capturing():
                keep = parts.get("keep") or ()
                for t in (*self._scatter, self._deg8, *(self._csr_in or ()), *(self._csr_out or ()), *(self._csr_in_packed or ()),
                          *[x for x in keep if torch.is_tensor(x)]):
                    if t is not None and t.is_cuda:
                        t.record_stream(side)
        return True

    # -- backward scatter CSR -------------------------------------------------
    @property
    def scatter(self):
        if self._scatter is None:
            self.build_hip()
        self._await()
        if self._scatter is None:
            dest = []
            for b in self.buckets:
                if b.count:
                    dest.append(torch.cat([b.sel.view(-1, 1), b.nei.view(b.count, b.degree)], dim=1).reshape(-1))
            if dest:
                dest = torch.cat(dest)
                order = torch.sort(dest, stable=True).indices
                counts = torch.bincount(dest, minlength=self.n_atoms)
            else:
                order = torch.zeros(0, dtype=torch.long, device=self.device)
                counts = torch.zeros(self.n_atoms, dtype=torch.long, device=self.device)
            rowptr = torch.zeros(self.n_atoms + 1, dtype=torch.int32, device=self.device)
            rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            self._scatter = (rowptr, order.to(torch.int32).contiguous())
        return self._scatter

    # -- propagate CSRs -------------------------------------------------------
    def _csr(self, key: torch.Tensor, val: torch.Tensor):
        order = torch.sort(key, stable=True).indices
        counts = torch.bincount(key, minlength=self.n_atoms)
        rowptr = torch.zeros(self.n_atoms + 1, dtype=torch.int32, device=key.device)
        rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        return rowptr, val[order].to(torch.int32).contiguous()

    @property
    def csr_in(self):
        """Edges grouped by target; columns are the sources (forward of propagate)."""
        if self._csr_in is None:
            self.build_hip()
        self._await()
        if self._csr_in is None:
            self._csr_in = self._csr(self.edge_index[1], self.edge_index[0])
        return self._csr_in

    @property
    def csr_out(self):
        """Edges grouped by source; columns are the targets (gradient of propagate)."""
        if self._csr_out is None:
            self.build_hip()
        self._await()
        if self._csr_out is None:
            self._csr_out = self._csr(self.edge_index[0], self.edge_index[1])
        return self._csr_out


    # -- block rows: which column block of a convolution output belongs to an atom -----
    @property
    def deg8(self):
        """[n_atoms] int8: the degree bucket every atom is in (0 = none)."""
        if self._deg8 is None:
            self.build_hip()
        self._await()
        if self._deg8 is None:
            d8 = torch.zeros(self.n_atoms, dtype=torch.int8, device=self.device)
            for b in self.buckets:
                if b.count:
                    d8[b.sel] = b.degree
            self._deg8 = d8
        return self._deg8

    @property
    def csr_in_packed(self):
        """``csr_in`` with the source atom's degree in bits 28..30 of every column entry
        (``mkgnn_segment_sum_block_rows`` mode 1)."""
        if self._csr_in_packed is None:
            self.build_hip()
        self._await()
        if self._csr_in_packed is None:
            rowptr, col = self.csr_in
            packed = col | (self.deg8[col.long()].to(torch.int32) << 28)
            self._csr_in_packed = (rowptr, packed.contiguous())
        return self._csr_in_packed

    def block_rows_ok(self, K: int) -> bool:
        """Can propagate run on block rows (``functional.propagate_add(..., blocks=...)``) for this batch?"""
        return (self.edge_index is not None and self.edge_index.numel() > 0 and self.n_atoms < (1 << 28) and 0 < K <= 255)


def plan_from_lists(n_atoms, p_focal_list, nei_p_list, nei_edge_attr_list, selected_index_list, nei_index_list,
                    edge_index=None, nei_edge_unit_list=None) -> BatchPlan:
    buckets = []
    for d in range(1, MAX_DEGREE + 1):
        buckets.append(Bucket(d, selected_index_list[d - 1], nei_index_list[d - 1], nei_edge_attr_list[d - 1],
                              p_focal_list[d - 1], nei_p_list[d - 1],
                              None if nei_edge_unit_list is None else nei_edge_unit_list[d - 1]))
    return BatchPlan(n_atoms, buckets, edge_index)


_PLAN_ATTR = "_mkgnn_plan"
_PLAN_CACHE: "dict" = {}
_PLAN_CACHE_MAX = 32


def plan_from_lists_cached(n_atoms, p_focal_list, nei_p_list, nei_edge_attr_list, selected_index_list, nei_index_list,
                           edge_index=None, nei_edge_unit_list=None, prebuilt=None) -> BatchPlan:
    """``plan_from_lists`` memoised on the identity of the index, bond-attribute and coordinate tensors (address,
    length, device, in-place version AND the tensor objects themselves, held weakly).

    ``MolGCN.forward`` receives the per-degree tensors as separate keyword arguments (the reference's
    signature, KernelLayer.py:53-87), so the batch object that would carry a cached plan is not
    visible there; resident batches keep their tensors alive, which makes the addresses a stable key.
    The plan's sorts run once per batch instead of once per step, and a step contains no
    host-synchronising call any more (required for hipGraph capture).
    """
    def ident(t):
        # address, length, device AND the in-place version counter: refilling a static input tensor in place (the usual
        # hipGraph pattern) changes the graph structure without changing the address
        return None if t is None else (t.data_ptr(), t.numel(), str(t.device), t._version)
    groups = (selected_index_list, nei_index_list, nei_edge_attr_list, p_focal_list, nei_p_list)
    tensors = [t for g in groups for t in g if t is not None] + ([edge_index] if edge_index is not None else [])
    key = (int(n_atoms),) + tuple(tuple(ident(t) for t in g) for g in groups) + (ident(edge_index),)
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        plan, refs = hit
        # addresses are recycled by the allocator: the entry only counts if these are the very tensor objects it was
        # built from (a freed tensor's weak reference is dead, a new tensor at the same address is another object)
        if len(refs) == len(tensors) and all(r() is t for r, t in zip(refs, tensors)):
            return plan
        del _PLAN_CACHE[key]
    plan = plan_from_lists(n_atoms, p_focal_list, nei_p_list, nei_edge_attr_list, selected_index_list,
                           nei_index_list, edge_index, nei_edge_unit_list)
    # build every index structure now: the lazy properties sort (and synchronise), which must not
    # happen inside a later backward pass or a hipGraph capture
    if prebuilt is not None and plan.adopt(prebuilt):
        pass                                 # (receptive_field.build_index_hip made them in the pass that made the buckets)
    elif not plan.build_hip():               # (on the index stream; joined by whoever reads a structure first)
        _ = plan.scatter
        if edge_index is not None:
            _ = plan.csr_in, plan.csr_out, plan.csr_in_packed
    for b in plan.buckets:
        if b.count and b.e_nei is not None and b.e_nei.is_cuda:
            b.e_unit(b.e_nei.numel() // (b.count * b.degree))
    if len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
        _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    _PLAN_CACHE[key] = (plan, [weakref.ref(t) for t in tensors])
    return plan


def plan_from_data(data) -> BatchPlan:
    """Plan for a reference-style ``data`` object, cached on the object itself."""
    cached = getattr(data, _PLAN_ATTR, None)
    fields = [getattr(data, f"{nm}_deg{d}") for nm in ("selected_index", "nei_index", "nei_edge_attr", "p_focal", "nei_p")
              for d in range(1, 5)]
    key = (data.x.shape[0], str(data.x.device)) + tuple((t.data_ptr(), t.numel(), t._version) for t in fields)
    if cached is not None and cached[0] == key:
        return cached[1]
    plan = plan_from_lists(
        data.x.shape[0],
        [getattr(data, f"p_focal_deg{d}") for d in range(1, 5)],
        [getattr(data, f"nei_p_deg{d}") for d in range(1, 5)],
        [getattr(data, f"nei_edge_attr_deg{d}") for d in range(1, 5)],
        [getattr(data, f"selected_index_deg{d}") for d in range(1, 5)],
        [getattr(data, f"nei_index_deg{d}") for d in range(1, 5)],
        getattr(data, "edge_index", None),
        [getattr(data, f"nei_edge_unit_deg{d}", None) for d in range(1, 5)])
    try:
        object.__setattr__(data, _PLAN_ATTR, (key, plan))
    except Exception:
        pass
    return plan
