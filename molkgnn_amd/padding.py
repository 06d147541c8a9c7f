"""Batches padded to one fixed shape, so that ONE captured hipGraph serves every batch of an epoch.

A hipGraph bakes every kernel's sizes in: atoms, atoms per degree, edges, molecules.  Batches of molecules differ in all
of them, so a step captured on one batch cannot replay another (bench.py's headline cycles resident batches, one graph
each).  Padding makes the sizes equal: every batch gets as many padding atoms of each degree as it is short of the
epoch's targets, dealt to ``PAD_MOLECULES`` (64) padding molecules and bonded among themselves in chains (the kernels read
indices, not chemistry: multiple bonds and self loops are fine there).  Nothing of a real molecule touches them -- the
edge list stays block diagonal -- and they cannot reach the loss:

* the node batch norm takes its statistics over the real atoms only (``n_valid_atoms``, read on the device:
  ``mkgnn_batchnorm_forward``'s ``n_valid_rows``);
* the padding molecules are the last rows of the graph embedding and are cut off before the head
  (``n_valid_molecules``), so their atoms receive a zero gradient and add exactly zero to every parameter gradient.

The step then runs from static buffers: ``StaticBatch.load`` copies the next padded batch in place, the captured graph
rebuilds degree buckets, unit bond rows and the index plan (``mkgnn_rf_*``, ``mkgnn_unit_rows8``, ``mkgnn_plan_build``:
no host synchronisation) and runs forward, backward and the optimiser.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .receptive_field import GraphBatch

RAW_FIELDS = ("x", "p", "edge_index", "edge_attr", "batch", "y")
PAD_MOLECULES = 64      # the padding atoms are dealt to this many padding molecules (a molecule is pooled by one wave: one
                        # molecule of a thousand atoms would take as long as 40 real ones)


def degree_histogram(batch: GraphBatch) -> List[int]:
    """[atoms, N_1, N_2, N_3, N_4, atoms in no bucket] of a collated batch (host values)."""
    n = batch.x.shape[0]
    deg = torch.bincount(batch.edge_index[0], minlength=n)
    h = torch.bincount(deg.clamp(max=5), minlength=6).tolist()
    return [n, h[1], h[2], h[3], h[4], h[0] + h[5]]


def fixed_shape(histograms: Sequence[Sequence[int]]) -> Dict[str, int]:
    """The common shape of a set of batches: per-degree targets = the maxima (+ 1 so that every batch receives a
    non-empty padding molecule, + the parity fix that makes the padding's bond stubs pair up)."""
    if any(h[5] for h in histograms):
        raise ValueError("atoms of degree 0 or > 4 are in no bucket: such batches cannot be padded to a bucket shape")
    t = [max(h[d] for h in histograms) + 1 for d in range(1, 5)]
    # sum_d d * (t_d - N_d) must be even for every batch; sum_d d * N_d = edges is even (two directed edges per bond),
    # so sum_d d * t_d must be even: one more degree-1 atom fixes an odd total
    if sum((d + 1) * t[d] for d in range(4)) % 2:
        t[0] += 1
    return {"n1": t[0], "n2": t[1], "n3": t[2], "n4": t[3], "atoms": sum(t), "edges": sum((d + 1) * t[d] for d in range(4))}


def pad_batch(batch: GraphBatch, shape: Dict[str, int], num_molecules: int) -> GraphBatch:
    """``batch`` (raw fields on any device, ``num_molecules`` molecules) + one padding molecule -> a batch of exactly
    ``shape``.  Returns the raw fields plus ``mol_ptr``, ``atom_mol`` (int32), ``n_valid_atoms`` (int64 scalar tensor),
    ``n_valid_molecules`` and ``bucket_sizes``."""
    dev = batch.x.device
    h = degree_histogram(batch)
    need = [shape[f"n{d}"] - h[d] for d in range(1, 5)]
    if min(need) < 0 or h[5]:
        raise ValueError(f"batch with degree histogram {h[1:5]} does not fit the shape {shape}")
    n_real, n_pad = h[0], sum(need)
    # padding atoms in degree order; their bond stubs paired off in sequence (stub k with stub k + 1)
    deg_of = torch.cat([torch.full((need[d],), d + 1, dtype=torch.long) for d in range(4)])
    stubs = torch.repeat_interleave(torch.arange(n_pad), deg_of) + n_real
    if stubs.numel() % 2:
        raise ValueError("odd number of padding bond stubs: the shape does not come from fixed_shape()")
    a, b = stubs[0::2], stubs[1::2]
    pad_ei = torch.stack([torch.stack([a, b], dim=1).reshape(-1), torch.stack([b, a], dim=1).reshape(-1)]).to(dev)
    E = batch.edge_attr.shape[1]
    pad_ea = torch.zeros(pad_ei.shape[1], E, dtype=batch.edge_attr.dtype, device=dev)
    pad_ea[:, 0] = 1.0                                   # (a valid bond vector: no zero-norm rows in the padding)
    out = GraphBatch()
    out.x = torch.cat([batch.x, torch.zeros(n_pad, batch.x.shape[1], dtype=batch.x.dtype, device=dev)])
    out.p = torch.cat([batch.p, torch.zeros(n_pad, batch.p.shape[1], dtype=batch.p.dtype, device=dev)])
    out.edge_index = torch.cat([batch.edge_index, pad_ei], dim=1)
    out.edge_attr = torch.cat([batch.edge_attr, pad_ea])
    # padding atoms -> PAD_MOLECULES padding molecules, contiguous runs (bonds between padding molecules are harmless: every
    # padding row of the embedding is cut off)
    pad_mol = num_molecules + (torch.arange(n_pad, device=dev) * PAD_MOLECULES // max(n_pad, 1)).to(batch.batch.dtype)
    out.batch = torch.cat([batch.batch, pad_mol])
    out.y = batch.y
    counts = torch.bincount(out.batch, minlength=num_molecules + PAD_MOLECULES)
    ptr = torch.zeros(num_molecules + PAD_MOLECULES + 1, dtype=torch.int32, device=dev)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    out.mol_ptr, out.atom_mol = ptr, out.batch.to(torch.int32)
    out.n_valid_atoms = torch.tensor([n_real], dtype=torch.int64, device=dev)
    out.n_valid_molecules, out.num_graphs = num_molecules, num_molecules + PAD_MOLECULES
    # the largest molecule of THIS batch, padding molecules included (atoms; directed edges leaving / entering it): what a
    # consumer that works molecule by molecule in on-chip memory -- readout.tail_loss -- has to know before it launches
    eb = out.batch[out.edge_index]
    out.max_mol_atoms = int(counts.max())
    out.max_mol_edges = int(max(torch.bincount(eb[0], minlength=1).max(), torch.bincount(eb[1], minlength=1).max()))
    out.bucket_sizes = [shape["n1"], shape["n2"], shape["n3"], shape["n4"]]
    assert out.x.shape[0] == shape["atoms"] and out.edge_index.shape[1] == shape["edges"]
    return out


def _layout(batch: GraphBatch, fields):
    """Byte offsets (256-aligned) of the fields inside one flat buffer."""
    off, table = 0, []
    for k in fields:
        t = getattr(batch, k)
        nbytes = t.numel() * t.element_size()
        table.append((k, off, tuple(t.shape), t.dtype, nbytes))
        off += (nbytes + 255) // 256 * 256
    return table, off


def pack(batch: GraphBatch) -> GraphBatch:
    """Move the fields of a padded batch into ONE flat buffer (``batch.flat``; the fields become views of it), so that
    loading it into the static buffers is a single device-to-device copy."""
    table, total = _layout(batch, StaticBatch.FIELDS)
    flat = torch.empty(total, dtype=torch.uint8, device=batch.x.device)
    for k, off, shape, dtype, nbytes in table:
        view = flat[off:off + nbytes].view(dtype).view(shape)
        view.copy_(getattr(batch, k))
        setattr(batch, k, view)
    batch.flat = flat
    return batch


class StaticBatch:
    """Static device buffers of one fixed shape: ``load`` copies a padded batch in place (the tensors keep their addresses,
    which is what a captured graph refers to); ``data`` is the object the model consumes.  The fields are views of one flat
    buffer: a batch prepared with ``pack`` is loaded by a single copy."""

    FIELDS = ("x", "p", "edge_index", "edge_attr", "batch", "y", "mol_ptr", "atom_mol", "n_valid_atoms")

    def __init__(self, first: GraphBatch, max_mol_atoms=None, max_mol_edges=None):
        """``max_mol_atoms`` / ``max_mol_edges``: the caller's bound on a molecule's atoms / directed edges for EVERY batch these
        buffers will hold (e.g. the maxima of ``pad_batch``'s figures over the epoch) -- the promise ``readout.tail_loss`` needs
        to run the fused tail in a step captured on them; without it the readout and the head stay separate operators."""
        self.table, total = _layout(first, self.FIELDS)
        self.flat = torch.empty(total, dtype=torch.uint8, device=first.x.device)
        self.data = GraphBatch()
        for k, off, shape, dtype, nbytes in self.table:
            view = self.flat[off:off + nbytes].view(dtype).view(shape)
            view.copy_(getattr(first, k))
            setattr(self.data, k, view)
        self.data.n_valid_molecules, self.data.num_graphs = first.n_valid_molecules, first.num_graphs
        self.data.bucket_sizes = list(first.bucket_sizes)
        # the molecule-size bound the consumer is shown (and a graph captured on these buffers is built for): the CALLER's promise
        # for every batch, never the first batch's own figures
        self.data.max_mol_atoms, self.data.max_mol_edges = max_mol_atoms, max_mol_edges

    def load(self, padded: GraphBatch) -> None:
        if list(padded.bucket_sizes) != self.data.bucket_sizes or padded.n_valid_molecules != self.data.n_valid_molecules:
            raise ValueError("batch shape differs from the static buffers'")
        for key in ("max_mol_atoms", "max_mol_edges"):
            promised, now = getattr(self.data, key, None), getattr(padded, key, None)
            if promised is not None and now is not None and now > promised:
                raise ValueError(f"{key} = {now} exceeds the bound {promised} these buffers were created with (a step captured on them "
                                 "may run the fused tail, readout.tail_loss, which relies on it)")
        flat = getattr(padded, "flat", None)
        if flat is not None and flat.numel() == self.flat.numel():
            self.flat.copy_(flat, non_blocking=True)
            return
        for k in self.FIELDS:
            getattr(self.data, k).copy_(getattr(padded, k), non_blocking=True)


class CompactStaticBatch:
    """Static device buffers fed in the compact wire form (``shards.compact_layout``; ``ShardLoader(compact=True)``):
    ``load`` copies a ``CompactBatch`` into the wire buffer, ``expand`` -- called INSIDE the captured step, before the
    receptive-field builder -- rebuilds ``edge_index``, ``edge_attr``, ``batch`` and ``atom_mol`` with one launch
    (``mkgnn_expand_batch``).  ``data`` is the object the model consumes: ``x``, ``p``, ``y``, ``mol_ptr`` and
    ``n_valid_atoms`` are views of the wire buffer, nothing is copied twice."""

    def __init__(self, shape: Dict[str, int], num_molecules: int, x_dim: int, p_dim: int, e_dim: int, device):
        import numpy as np
        from .shards import compact_layout
        self.shape, self.num_molecules, self.e_dim = dict(shape), int(num_molecules), int(e_dim)
        self.table, total = compact_layout(shape, num_molecules, x_dim, p_dim, e_dim)
        dev = torch.device(device)
        self.wire = torch.zeros(total, dtype=torch.uint8, device=dev)
        tdt = {np.float32: torch.float32, np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8}
        v = {k: self.wire[off:off + nbytes].view(tdt[dt]).view(shp) for k, off, shp, dt, nbytes in self.table}
        self._v = v
        A, E2 = shape["atoms"], shape["edges"]
        self.data = GraphBatch(
            x=v["x"], p=v["p"], y=v["y"], mol_ptr=v["mol_ptr"], n_valid_atoms=v["n_valid_atoms"],
            edge_index=torch.zeros((2, E2), dtype=torch.int64, device=dev),
            edge_attr=torch.zeros((E2, e_dim), dtype=torch.float32, device=dev),
            batch=torch.zeros(A, dtype=torch.int64, device=dev), atom_mol=torch.zeros(A, dtype=torch.int32, device=dev))
        self.data.n_valid_molecules, self.data.num_graphs = self.num_molecules, self.num_molecules + PAD_MOLECULES
        self.data.bucket_sizes = [shape["n1"], shape["n2"], shape["n3"], shape["n4"]]

    def load(self, batch) -> None:
        if list(batch.bucket_sizes) != self.data.bucket_sizes or batch.n_valid_molecules != self.num_molecules \
                or batch.flat.numel() != self.wire.numel():
            raise ValueError("batch shape differs from the static buffers'")
        self.wire.copy_(batch.flat, non_blocking=True)

    def expand(self) -> None:
        from . import _lib
        d, v = self.data, self._v
        with torch.cuda.device(self.wire.device):
            _lib.check(_lib.load().mkgnn_expand_batch(
                v["bond_ij"].data_ptr(), v["bond_attr"].data_ptr(), v["bond_ij"].shape[0], self.e_dim, v["mol_ptr"].data_ptr(),
                d.num_graphs, d.x.shape[0], d.edge_index.data_ptr(), d.edge_attr.data_ptr(), d.batch.data_ptr(),
                d.atom_mol.data_ptr(), _lib.stream_ptr(self.wire.device)), "mkgnn_expand_batch")
