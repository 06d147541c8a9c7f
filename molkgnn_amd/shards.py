"""Packed molecule shards: the pre-collated on-disk / in-memory format of SURVEY.md 8 f-2.

The reference stores one PyG ``Data`` object per molecule (``wrapper.py:362-460``, ``InMemoryDataset``) and lets
``DataLoader`` workers collate a batch from B Python objects with 20 extra per-degree tensors each
(``wrapper.py:559-672``).  At the ≥ 10^6 molecules/s the kernels run at, that loader is the bound.  A shard keeps the
molecules of a (pre-shuffled) part of the data set **already concatenated**, in the order they will be consumed:

* a batch is a contiguous molecule range, so collation is five slice copies into one pinned staging buffer and ONE
  host-to-device copy (the slices of ``x``, ``p``, ``edge_attr``, the bond endpoints and the molecule pointers);
* everything index-like is rebuilt on the device: batch-local atom ids (stored ids are shard-global int32: one
  subtraction), the ``batch`` vector, and then the degree buckets / index plan by the HIP builders
  (``receptive_field.attach_receptive_fields``, ``plan.BatchPlan.build_hip``);
* the file is a fixed header + 64-byte aligned little-endian arrays, read through ``numpy.memmap`` (no parsing, no
  pickling; a shard written on one box is mapped on another);
* data-parallel ranks take batches ``rank, rank + world, ...`` of the same shard sequence, so the union over
  ranks is the single-GPU stream.

Layout (version 2)::

    header   256 bytes: b"MKGS", u32 version, u64 n_molecules, u64 n_atoms, u64 n_bonds (directed edges),
                        u32 x_dim, u32 e_dim, u32 p_dim, u32 flags (bit 0: the compact wire form applies), then 10 x u64
                        array offsets in the order below
    mol_atom_ptr  i64 [M + 1]     first atom of every molecule
    mol_edge_ptr  i64 [M + 1]     first directed edge of every molecule
    y             f32 [M]         label (data.py:37 trains on one task)
    assay_id      i32 [M]         PubChem AID the molecule comes from (mixed nine-assay shards, BASELINE configs[4])
    x             f32 [A, x_dim]
    p             f32 [A, p_dim]
    edge_src      i32 [E]         shard-global atom ids; bonds as consecutive (i, j), (j, i) (wrapper.py:152-156)
    edge_dst      i32 [E]
    edge_attr     f32 [E, e_dim]
    mol_deg_ptr   i64 [M + 1, 5]  prefix sums over molecules of their atoms of degree 1, 2, 3, 4 and of any other degree:
                                  the degree histogram of a molecule range is one subtraction (fixed-shape padding)

Fixed-shape batches (``padding.py``: one captured graph serves every batch) come straight out of the loader:
``ShardLoader(..., fixed_shape=True)`` pads every batch ON THE HOST while staging it -- the padding atoms, their bonds
and molecules exactly as ``padding.pad_batch`` makes them -- directly in the layout of ``padding.StaticBatch``'s flat
buffer, so a batch reaches the static buffers by one host-to-device copy and one device copy, with no device-side
index work and no host synchronisation.
"""
from __future__ import annotations

import atexit
import os
import struct
import weakref
from queue import Queue
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .receptive_field import GraphBatch

MAGIC = b"MKGS"
VERSION = 2
HEADER_BYTES = 256
ALIGN = 64
_ARRAYS = ("mol_atom_ptr", "mol_edge_ptr", "y", "assay_id", "x", "p", "edge_src", "edge_dst", "edge_attr", "mol_deg_ptr")
_HEAD = struct.Struct("<4sIQQQIIII10Q")


def _align(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


def write_shard(path: str, batch: GraphBatch) -> None:
    """Write the molecules of a collated batch (``x``, ``p``, ``edge_index``, ``edge_attr``, ``batch``, ``y`` and
    optionally ``assay_id``) as one shard, in the batch's molecule order."""
    x = batch.x.detach().cpu().numpy().astype(np.float32, copy=False)
    p = batch.p.detach().cpu().numpy().astype(np.float32, copy=False)
    ei = batch.edge_index.detach().cpu().numpy()
    ea = batch.edge_attr.detach().cpu().numpy().astype(np.float32, copy=False)
    bvec = batch.batch.detach().cpu().numpy()
    y = batch.y.detach().cpu().numpy().astype(np.float32, copy=False).reshape(-1)
    m = int(y.shape[0])
    a, e = int(x.shape[0]), int(ei.shape[1])
    if a >= 2 ** 31 or e >= 2 ** 31:
        raise ValueError("a shard holds fewer than 2^31 atoms and directed edges (ids are int32)")
    if bvec.shape[0] != a or (a and (np.diff(bvec) < 0).any()):
        raise ValueError("`batch` must be sorted (atoms grouped by molecule), one entry per atom")
    edge_mol = bvec[ei[0]] if e else np.zeros(0, dtype=np.int64)
    if e and ((np.diff(edge_mol) < 0).any() or (bvec[ei[1]] != edge_mol).any()):
        raise ValueError("edges must be grouped by molecule and stay inside it")
    atom_ptr = np.zeros(m + 1, dtype=np.int64)
    atom_ptr[1:] = np.cumsum(np.bincount(bvec, minlength=m))
    edge_ptr = np.zeros(m + 1, dtype=np.int64)
    edge_ptr[1:] = np.cumsum(np.bincount(edge_mol, minlength=m))
    assay = getattr(batch, "assay_id", None)
    assay = np.zeros(m, dtype=np.int32) if assay is None else assay.detach().cpu().numpy().astype(np.int32)
    deg = np.bincount(ei[0], minlength=a) if e else np.zeros(a, dtype=np.int64)
    cls = np.where((deg >= 1) & (deg <= 4), deg - 1, 4)                 # 0..3: degree 1..4, 4: in no bucket
    per_mol = np.zeros((m, 5), dtype=np.int64)
    np.add.at(per_mol, (bvec, cls), 1)
    deg_ptr = np.zeros((m + 1, 5), dtype=np.int64)
    deg_ptr[1:] = np.cumsum(per_mol, axis=0)
    arrays = {"mol_atom_ptr": atom_ptr, "mol_edge_ptr": edge_ptr, "y": y, "assay_id": assay,
              "x": np.ascontiguousarray(x), "p": np.ascontiguousarray(p),
              "edge_src": ei[0].astype(np.int32), "edge_dst": ei[1].astype(np.int32), "edge_attr": np.ascontiguousarray(ea),
              "mol_deg_ptr": deg_ptr}
    offsets, off = [], HEADER_BYTES
    for name in _ARRAYS:
        offsets.append(off)
        off = _align(off + arrays[name].nbytes)
    # can the batches of this shard travel in the compact wire form (collate_compact)?  Bonds stored as consecutive
    # (i, j), (j, i) pairs with one attribute row for both directions, attributes byte-valued
    compact = e % 2 == 0 and (e == 0 or (
        np.array_equal(ei[0, 0::2], ei[1, 1::2]) and np.array_equal(ei[1, 0::2], ei[0, 1::2]) and
        np.array_equal(ea[0::2], ea[1::2]) and float(ea.min()) >= 0.0 and float(ea.max()) <= 255.0 and
        np.array_equal(ea, np.rint(ea))))
    head = _HEAD.pack(MAGIC, VERSION, m, a, e, x.shape[1], ea.shape[1] if ea.ndim == 2 else 0, p.shape[1], 1 if compact else 0,
                      *offsets)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(head.ljust(HEADER_BYTES, b"\0"))
        for name, o in zip(_ARRAYS, offsets):
            f.seek(o)
            f.write(arrays[name].tobytes())
        f.truncate(off)
    os.replace(tmp, path)


class Shard:
    """A memory-mapped shard; ``slices(m0, m1)`` are zero-copy views of a molecule range."""

    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as f:
            head = f.read(HEADER_BYTES)
        if len(head) < _HEAD.size:
            raise ValueError(f"{path}: not a molecule shard (short header)")
        magic, version, m, a, e, xd, ed, pd, flags, *offs = _HEAD.unpack(head[:_HEAD.size])
        if magic != MAGIC:
            raise ValueError(f"{path}: not a molecule shard (magic {magic!r})")
        if version != VERSION:
            raise ValueError(f"{path}: shard version {version}, this reader knows {VERSION}")
        self.n_molecules, self.n_atoms, self.n_edges = int(m), int(a), int(e)
        self.x_dim, self.e_dim, self.p_dim = int(xd), int(ed), int(pd)
        self.compact_ok = bool(flags & 1)    # bonds as reversed pairs with shared, byte-valued attributes: collate_compact applies
        shapes = {"mol_atom_ptr": (np.int64, (m + 1,)), "mol_edge_ptr": (np.int64, (m + 1,)), "y": (np.float32, (m,)),
                  "assay_id": (np.int32, (m,)), "x": (np.float32, (a, xd)), "p": (np.float32, (a, pd)),
                  "edge_src": (np.int32, (e,)), "edge_dst": (np.int32, (e,)), "edge_attr": (np.float32, (e, ed)),
                  "mol_deg_ptr": (np.int64, (m + 1, 5))}
        size = os.path.getsize(path)
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        for name, o in zip(_ARRAYS, offs):
            dt, shp = shapes[name]
            nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
            if o % ALIGN or o + nbytes > size:
                raise ValueError(f"{path}: array {name} at {o} (+{nbytes}) does not fit the file ({size} bytes)")
            setattr(self, name, np.frombuffer(self._mm, dtype=dt, count=int(np.prod(shp)), offset=int(o)).reshape(shp))
        if self.mol_atom_ptr[-1] != a or self.mol_edge_ptr[-1] != e:
            raise ValueError(f"{path}: molecule pointers do not cover the arrays")

    def degree_histogram(self, m0: int, m1: int) -> List[int]:
        """``padding.degree_histogram`` of the molecules [m0, m1): [atoms, N_1, N_2, N_3, N_4, atoms in no bucket]."""
        d = (self.mol_deg_ptr[m1] - self.mol_deg_ptr[m0]).tolist()
        return [int(self.mol_atom_ptr[m1] - self.mol_atom_ptr[m0]), d[0], d[1], d[2], d[3], d[4]]

    def ranges(self, m0: int, m1: int):
        return (int(self.mol_atom_ptr[m0]), int(self.mol_atom_ptr[m1]), int(self.mol_edge_ptr[m0]), int(self.mol_edge_ptr[m1]))


def _stage_layout(n_mol: int, n_atoms: int, n_edges: int, xd: int, pd: int, ed: int):
    """Byte offsets of a batch's slices in the staging buffer (each 64-byte aligned) and the total."""
    parts = (("x", 4 * n_atoms * xd), ("p", 4 * n_atoms * pd), ("edge_attr", 4 * n_edges * ed), ("edge_src", 4 * n_edges),
             ("edge_dst", 4 * n_edges), ("atom_ptr", 8 * (n_mol + 1)), ("y", 4 * n_mol), ("assay_id", 4 * n_mol))
    out, off = {}, 0
    for name, nbytes in parts:
        out[name] = (off, nbytes)
        off = _align(off + nbytes)
    return out, off


def collate(shard: Shard, m0: int, m1: int, device, staging: Optional[torch.Tensor] = None,
            stream: Optional["torch.cuda.Stream"] = None) -> GraphBatch:
    """Molecules ``[m0, m1)`` of a shard as one collated batch on ``device``: slice copies into ONE staging buffer
    (pinned when the device is a GPU), one host-to-device copy, index arithmetic on the device."""
    device = torch.device(device)
    a0, a1, e0, e1 = shard.ranges(m0, m1)
    nm, na, ne = m1 - m0, a1 - a0, e1 - e0
    lay, total = _stage_layout(nm, na, ne, shard.x_dim, shard.p_dim, shard.e_dim)
    if staging is None or staging.numel() < total:
        staging = torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=(device.type == "cuda"))
    host = staging.numpy()

    def put(name, arr):
        o, nbytes = lay[name]
        host[o:o + nbytes] = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)

    put("x", shard.x[a0:a1]); put("p", shard.p[a0:a1]); put("edge_attr", shard.edge_attr[e0:e1])
    put("edge_src", shard.edge_src[e0:e1]); put("edge_dst", shard.edge_dst[e0:e1])
    put("atom_ptr", shard.mol_atom_ptr[m0:m1 + 1]); put("y", shard.y[m0:m1]); put("assay_id", shard.assay_id[m0:m1])
    if device.type == "cuda":
        with torch.cuda.stream(stream) if stream is not None else _null():
            dev = staging[:total].to(device, non_blocking=True)
    else:
        dev = staging[:total].clone()

    def view(name, dtype, shape):
        o, nbytes = lay[name]
        return dev[o:o + nbytes].view(dtype).view(shape)

    with torch.cuda.stream(stream) if (stream is not None and device.type == "cuda") else _null():
        src = view("edge_src", torch.int32, (ne,)).long() - a0
        dst = view("edge_dst", torch.int32, (ne,)).long() - a0
        atom_ptr = view("atom_ptr", torch.int64, (nm + 1,)) - a0
        counts = atom_ptr[1:] - atom_ptr[:-1]
        batch_vec = torch.repeat_interleave(torch.arange(nm, device=device), counts, output_size=na)
        out = GraphBatch(x=view("x", torch.float32, (na, shard.x_dim)), p=view("p", torch.float32, (na, shard.p_dim)),
                         edge_index=torch.stack([src, dst]), edge_attr=view("edge_attr", torch.float32, (ne, shard.e_dim)),
                         batch=batch_vec, y=view("y", torch.float32, (nm,)), num_graphs=nm, smiles=None,
                         assay_id=view("assay_id", torch.int32, (nm,)).long(), mol_ptr=atom_ptr)
    out._staging = staging            # (the pinned buffer must outlive the asynchronous copy)
    return out


def padded_layout(shape, num_molecules: int, x_dim: int, p_dim: int, e_dim: int):
    """Field table of ``padding.StaticBatch``'s flat buffer for a fixed shape: (name, offset, shape, numpy dtype, bytes)."""
    from .padding import PAD_MOLECULES, StaticBatch
    A, Eg, G = shape["atoms"], shape["edges"], num_molecules + PAD_MOLECULES
    spec = {"x": ((A, x_dim), np.float32), "p": ((A, p_dim), np.float32), "edge_index": ((2, Eg), np.int64),
            "edge_attr": ((Eg, e_dim), np.float32), "batch": ((A,), np.int64), "y": ((num_molecules,), np.float32),
            "mol_ptr": ((G + 1,), np.int32), "atom_mol": ((A,), np.int32), "n_valid_atoms": ((1,), np.int64)}
    table, off = [], 0
    for k in StaticBatch.FIELDS:
        shp, dt = spec[k]
        nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
        table.append((k, off, shp, dt, nbytes))
        off += (nbytes + 255) // 256 * 256
    return table, off


def collate_padded(shard: Shard, m0: int, m1: int, shape, host: np.ndarray) -> None:
    """Molecules ``[m0, m1)`` padded to ``shape`` (``padding.fixed_shape``), written into ``host`` (uint8, the flat
    layout of ``padded_layout``): what ``padding.pack(padding.pad_batch(collate(...), shape, m1 - m0))`` holds, made on
    the host with slice copies and a few small index fills."""
    from .padding import PAD_MOLECULES
    nm = m1 - m0
    a0, a1, e0, e1 = shard.ranges(m0, m1)
    na, ne = a1 - a0, e1 - e0
    h = shard.degree_histogram(m0, m1)
    need = [shape[f"n{d}"] - h[d] for d in range(1, 5)]
    if min(need) < 0 or h[5]:
        raise ValueError(f"molecules [{m0}, {m1}) with degree histogram {h[1:5]} (+{h[5]} in no bucket) do not fit the shape {shape}")
    n_pad = sum(need)
    table, total = padded_layout(shape, nm, shard.x_dim, shard.p_dim, shard.e_dim)
    if host.shape[0] < total:
        raise ValueError("staging buffer too small")
    f = {k: host[off:off + nbytes].view(dt).reshape(shp) for k, off, shp, dt, nbytes in table}
    f["x"][:na] = shard.x[a0:a1]; f["x"][na:] = 0.0
    f["p"][:na] = shard.p[a0:a1]; f["p"][na:] = 0.0
    f["edge_attr"][:ne] = shard.edge_attr[e0:e1]
    f["edge_attr"][ne:] = 0.0
    f["edge_attr"][ne:, 0] = 1.0
    ei = f["edge_index"]
    np.subtract(shard.edge_src[e0:e1], a0, out=ei[0, :ne], casting="unsafe")
    np.subtract(shard.edge_dst[e0:e1], a0, out=ei[1, :ne], casting="unsafe")
    # padding atoms in degree order, their bond stubs paired off in sequence (padding.pad_batch)
    deg_of = np.repeat(np.arange(1, 5), need)
    stubs = np.repeat(np.arange(n_pad, dtype=np.int64), deg_of) + na
    if stubs.shape[0] % 2 or stubs.shape[0] != shape["edges"] - ne:
        raise ValueError("padding bond stubs do not pair up: the shape does not come from fixed_shape() over these batches")
    a, b = stubs[0::2], stubs[1::2]
    ei[0, ne::2] = a; ei[0, ne + 1::2] = b
    ei[1, ne::2] = b; ei[1, ne + 1::2] = a
    atom_ptr = shard.mol_atom_ptr[m0:m1 + 1] - a0
    counts = np.diff(atom_ptr)
    bt = f["batch"]
    bt[:na] = np.repeat(np.arange(nm, dtype=np.int64), counts)
    bt[na:] = nm + (np.arange(n_pad, dtype=np.int64) * PAD_MOLECULES) // max(n_pad, 1)
    f["atom_mol"][:] = bt
    f["y"][:] = shard.y[m0:m1]
    mp = f["mol_ptr"]
    mp[0] = 0
    mp[1:nm + 1] = atom_ptr[1:]
    pad_counts = np.bincount(bt[na:] - nm, minlength=PAD_MOLECULES)
    mp[nm + 1:] = na + np.cumsum(pad_counts)
    f["n_valid_atoms"][0] = na


def compact_layout(shape, num_molecules: int, x_dim: int, p_dim: int, e_dim: int):
    """Field table of the compact wire form of a fixed-shape batch (``padding.CompactStaticBatch``): what has to cross
    PCIe and nothing derived -- features and coordinates as they are, every bond once as an int32 pair with byte-valued
    attributes, labels, molecule pointers; ``mkgnn_expand_batch`` rebuilds ``edge_index`` (int64, both directions),
    ``edge_attr`` (fp32, both directions), ``batch`` and ``atom_mol`` on the device.  14.5 MB instead of 23.8 MB for 4096
    molecules."""
    from .padding import PAD_MOLECULES
    A, B2, G = shape["atoms"], shape["edges"] // 2, num_molecules + PAD_MOLECULES
    spec = (("x", (A, x_dim), np.float32), ("p", (A, p_dim), np.float32), ("bond_ij", (B2, 2), np.int32),
            ("bond_attr", (B2, e_dim), np.uint8), ("y", (num_molecules,), np.float32), ("mol_ptr", (G + 1,), np.int32),
            ("n_valid_atoms", (1,), np.int64))
    table, off = [], 0
    for k, shp, dt in spec:
        nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
        table.append((k, off, shp, dt, nbytes))
        off += (nbytes + 255) // 256 * 256
    return table, off


def collate_compact(shard: Shard, m0: int, m1: int, shape, host: np.ndarray) -> None:
    """Molecules ``[m0, m1)`` padded to ``shape`` in the compact wire form (``compact_layout``); expands (``mkgnn_expand_batch``)
    to exactly what ``collate_padded`` writes.  Needs ``shard.compact_ok``."""
    from .padding import PAD_MOLECULES
    if not shard.compact_ok:
        raise ValueError(f"{shard.path}: bonds are not reversed pairs with shared byte-valued attributes: no compact form")
    nm = m1 - m0
    a0, a1, e0, e1 = shard.ranges(m0, m1)
    na, ne = a1 - a0, e1 - e0
    h = shard.degree_histogram(m0, m1)
    need = [shape[f"n{d}"] - h[d] for d in range(1, 5)]
    if min(need) < 0 or h[5]:
        raise ValueError(f"molecules [{m0}, {m1}) with degree histogram {h[1:5]} (+{h[5]} in no bucket) do not fit the shape {shape}")
    n_pad = sum(need)
    table, total = compact_layout(shape, nm, shard.x_dim, shard.p_dim, shard.e_dim)
    if host.shape[0] < total:
        raise ValueError("staging buffer too small")
    f = {k: host[off:off + nbytes].view(dt).reshape(shp) for k, off, shp, dt, nbytes in table}
    nb = ne // 2
    f["x"][:na] = shard.x[a0:a1]; f["x"][na:] = 0.0
    f["p"][:na] = shard.p[a0:a1]; f["p"][na:] = 0.0
    f["bond_attr"][:nb] = shard.edge_attr[e0:e1:2]           # (float -> uint8: exact, the shard checked the values)
    f["bond_attr"][nb:] = 0
    f["bond_attr"][nb:, 0] = 1
    ij = f["bond_ij"]
    np.subtract(shard.edge_src[e0:e1:2], a0, out=ij[:nb, 0], casting="unsafe")
    np.subtract(shard.edge_dst[e0:e1:2], a0, out=ij[:nb, 1], casting="unsafe")
    deg_of = np.repeat(np.arange(1, 5), need)
    stubs = np.repeat(np.arange(n_pad, dtype=np.int64), deg_of) + na
    if stubs.shape[0] % 2 or stubs.shape[0] != shape["edges"] - ne:
        raise ValueError("padding bond stubs do not pair up: the shape does not come from fixed_shape() over these batches")
    ij[nb:, 0] = stubs[0::2]
    ij[nb:, 1] = stubs[1::2]
    atom_ptr = shard.mol_atom_ptr[m0:m1 + 1] - a0
    f["y"][:] = shard.y[m0:m1]
    mp = f["mol_ptr"]
    mp[0] = 0
    mp[1:nm + 1] = atom_ptr[1:]
    pad_mol = (np.arange(n_pad, dtype=np.int64) * PAD_MOLECULES) // max(n_pad, 1)
    mp[nm + 1:] = na + np.cumsum(np.bincount(pad_mol, minlength=PAD_MOLECULES))
    f["n_valid_atoms"][0] = na


def collate_compact_native(shard: Shard, m0: int, m1: int, shape, host: np.ndarray) -> None:
    """``collate_compact`` by ONE call into the library (``mkgnn_collate_compact``, host code): byte for byte the same buffer
    (tested), and the interpreter lock is released for the whole batch -- the numpy form holds it through a dozen small array
    operations per batch, which with three loader threads kept the thread that replays the training graph waiting (0.2 ms of
    idle GPU per step of a shard-fed epoch)."""
    from . import _lib
    from .padding import PAD_MOLECULES
    lib = _lib.load()
    view = getattr(shard, "_view", None)
    if view is None:
        if not shard.compact_ok:
            raise ValueError(f"{shard.path}: bonds are not reversed pairs with shared byte-valued attributes: no compact form")
        view = _lib.ShardView()
        for name in ("x", "p", "edge_src", "edge_dst", "edge_attr", "y", "mol_atom_ptr", "mol_edge_ptr", "mol_deg_ptr"):
            setattr(view, name, getattr(shard, name).ctypes.data)
        view.n_molecules, view.x_dim, view.p_dim, view.e_dim = shard.n_molecules, shard.x_dim, shard.p_dim, shard.e_dim
        shard._view = view
    sh = _lib.Int64x6(shape["atoms"], shape["edges"], shape["n1"], shape["n2"], shape["n3"], shape["n4"])
    _lib.check(lib.mkgnn_collate_compact(view, int(m0), int(m1), sh, PAD_MOLECULES, host.ctypes.data, host.shape[0]),
               "mkgnn_collate_compact")


class CompactBatch:
    """A fixed-shape batch in the compact wire form on the device (``padding.CompactStaticBatch.load`` takes it)."""

    def __init__(self, flat, shape, num_molecules: int):
        self.flat, self.shape, self.n_valid_molecules = flat, dict(shape), num_molecules
        self.bucket_sizes = [shape["n1"], shape["n2"], shape["n3"], shape["n4"]]


class PackedBatch:
    """A fixed-shape batch as one flat device buffer in ``padding.StaticBatch``'s layout (``StaticBatch.load`` takes it)."""

    def __init__(self, flat, shape, num_molecules: int):
        from .padding import PAD_MOLECULES
        self.flat = flat
        self.bucket_sizes = [shape["n1"], shape["n2"], shape["n3"], shape["n4"]]
        self.n_valid_molecules, self.num_graphs = num_molecules, num_molecules + PAD_MOLECULES

    def unpack(self, shard_dims) -> GraphBatch:
        """The fields as views of the flat buffer (a ``padding.pad_batch`` result)."""
        x_dim, p_dim, e_dim = shard_dims
        shape = {"n1": self.bucket_sizes[0], "n2": self.bucket_sizes[1], "n3": self.bucket_sizes[2], "n4": self.bucket_sizes[3]}
        shape["atoms"] = sum(self.bucket_sizes)
        shape["edges"] = sum((d + 1) * self.bucket_sizes[d] for d in range(4))
        table, _ = padded_layout(shape, self.n_valid_molecules, x_dim, p_dim, e_dim)
        out = GraphBatch()
        tdt = {np.float32: torch.float32, np.int64: torch.int64, np.int32: torch.int32}
        for k, off, shp, dt, nbytes in table:
            setattr(out, k, self.flat[off:off + nbytes].view(tdt[dt]).view(shp))
        out.flat = self.flat
        out.bucket_sizes, out.n_valid_molecules, out.num_graphs = self.bucket_sizes, self.n_valid_molecules, self.num_graphs
        return out


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_LOADERS: "weakref.WeakSet" = weakref.WeakSet()


def _release_all() -> None:
    """Pinned staging and device landing buffers of loaders that are still alive go BEFORE the interpreter tears the GPU
    runtime down (a pinned buffer freed after that aborts the process)."""
    for ld in list(_LOADERS):
        ld.close()


atexit.register(_release_all)


class ShardLoader:
    """Batches of ``batch_size`` consecutive molecules from a sequence of shards, for rank ``rank`` of ``world``.

    Batch k of the global sequence (shard after shard, a shard's tail batch may be short) goes to rank ``k % world``
    (SURVEY 8e).  ``workers`` background threads stage the next batches into pinned buffers and issue their
    host-to-device copies on a copy stream, ``prefetch`` batches ahead, handed over in order; the consumer's stream waits
    for the copy's event, not for the host.
    """

    def __init__(self, paths: Sequence[str], batch_size: int, device="cpu", rank: int = 0, world: int = 1,
                 prefetch: int = 2, drop_last: bool = False, workers: int = 2, fixed_shape: bool = False,
                 compact: bool = False):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} of world {world}")
        self.paths, self.batch_size, self.device = list(paths), int(batch_size), torch.device(device)
        self.rank, self.world, self.prefetch, self.drop_last = rank, world, max(1, int(prefetch)), drop_last
        self.workers = max(1, int(workers))
        self.shards = [Shard(p) for p in self.paths]
        # fixed_shape: every batch padded on the host to the common shape of this rank's batches (padding.fixed_shape) and
        # handed over as a PackedBatch; short tail batches are dropped (the static buffers hold exactly batch_size molecules)
        self._pinned: list = []                          # (pinned staging buffer, event of its last host-to-device copy)
        self._landing: list = []
        self._consumed: list = []                        # per landing buffer: the consumer's stream has passed its last use
        self._copy_stream = None
        _LOADERS.add(self)
        self.shape = None
        # compact (with fixed_shape): batches travel in the compact wire form (collate_compact) and come out as CompactBatch
        self.compact = bool(compact)
        if self.compact and not fixed_shape:
            raise ValueError("the compact wire form is a fixed-shape form: fixed_shape=True is required")
        if self.compact and not all(sh.compact_ok for sh in self.shards):
            raise ValueError("a shard's bonds do not allow the compact wire form")
        if fixed_shape:
            from .padding import fixed_shape as _fixed_shape
            self.drop_last = True
            work = self.plan()
            if not work:
                raise ValueError("no full batch for this rank")
            self.shape = _fixed_shape([self.shards[si].degree_histogram(m0, m1) for si, m0, m1 in work])

    def close(self) -> None:
        """Drop the pinned staging and device landing buffers (they are re-made on the next epoch)."""
        if self._copy_stream is not None:
            try:
                self._copy_stream.synchronize()          # no copy may still be reading a pinned buffer that is about to go
            except Exception:                            # (interpreter shutdown: the runtime may be gone already)
                pass
        self._pinned, self._landing, self._consumed = [], [], []

    def plan(self) -> List[tuple]:
        """``(shard index, m0, m1)`` of this rank's batches, in order."""
        out, k = [], 0
        for si, sh in enumerate(self.shards):
            for m0 in range(0, sh.n_molecules, self.batch_size):
                m1 = min(m0 + self.batch_size, sh.n_molecules)
                if self.drop_last and m1 - m0 < self.batch_size:
                    continue
                if k % self.world == self.rank:
                    out.append((si, m0, m1))
                k += 1
        return out

    def __len__(self):
        return len(self.plan())

    def __iter__(self) -> Iterator[GraphBatch]:
        work = self.plan()
        if self.device.type != "cuda":
            for si, m0, m1 in work:
                if self.shape is None:
                    yield collate(self.shards[si], m0, m1, self.device)
                else:
                    sh = self.shards[si]
                    layout, fill, make = (compact_layout, collate_compact, CompactBatch) if self.compact else \
                        (padded_layout, collate_padded, PackedBatch)
                    _, total = layout(self.shape, m1 - m0, sh.x_dim, sh.p_dim, sh.e_dim)
                    flat = torch.empty(total, dtype=torch.uint8)
                    fill(sh, m0, m1, self.shape, flat.numpy())
                    yield make(flat, self.shape, m1 - m0)
            return
        from concurrent.futures import ThreadPoolExecutor
        # One copy stream for the loader's life, and every buffer keeps the event that guards its reuse ACROSS epochs: a
        # pinned staging buffer the event of its last host-to-device copy, a landing buffer the event the consumer's stream
        # recorded after its last use.  (A fresh stream and forgotten events per epoch let a new epoch's workers overwrite
        # buffers whose copies -- or whose consumer -- of the previous epoch's tail had not run yet whenever the GPU lagged
        # the host by a batch or more.)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        copy_stream = self._copy_stream
        depth = self.prefetch + self.workers
        free: Queue = Queue()                            # pinned staging buffers with the event of their last copy
        for k in range(depth + 1):                       # (kept across epochs: pinning memory costs milliseconds)
            free.put(self._pinned[k] if k < len(self._pinned) else (None, None))
        self._pinned = []
        # fixed-shape batches land in a ring of device buffers that is allocated once: a fresh device tensor per batch
        # would have the allocator wait for (or grow past) blocks the consumer's stream has not released yet.  A landing
        # buffer is reused only after the consumer's stream has passed the use of the batch it held (event recorded when
        # the consumer asks for the next batch), so a batch's ``flat`` is valid until the loader has handed out
        # ``prefetch + workers + 1`` more batches.
        K = depth + 2
        if len(self._landing) != K:
            self._landing = [None] * K
            self._consumed = [None] * K
        consumed = self._consumed

        def stage(i, si, m0, m1):
            staging, last = free.get()
            if last is not None:
                last.synchronize()                       # the buffer's previous copy has left the host
            with torch.cuda.device(self.device):
                if self.shape is None:
                    b = collate(self.shards[si], m0, m1, self.device, staging, copy_stream)
                else:
                    sh = self.shards[si]
                    layout, fill, make = (compact_layout, collate_compact_native, CompactBatch) if self.compact else \
                        (padded_layout, collate_padded, PackedBatch)
                    _, total = layout(self.shape, m1 - m0, sh.x_dim, sh.p_dim, sh.e_dim)
                    if staging is None or staging.numel() < total:
                        staging = torch.empty(total, dtype=torch.uint8, pin_memory=True)
                    fill(sh, m0, m1, self.shape, staging.numpy())
                    k = i % K
                    if self._landing[k] is None or self._landing[k].numel() < total:
                        self._landing[k] = torch.empty(total, dtype=torch.uint8, device=self.device)
                    with torch.cuda.stream(copy_stream):
                        if consumed[k] is not None:
                            copy_stream.wait_event(consumed[k])
                        dst = self._landing[k][:total]
                        dst.copy_(staging[:total], non_blocking=True)
                        b = make(dst, self.shape, m1 - m0)
                    b._staging, b._slot = staging, k
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return b, ev

        with ThreadPoolExecutor(self.workers) as pool:   # (numpy's slice copies release the GIL: the workers overlap)
            pending = []
            it = iter(enumerate(work))
            for i, w in it:
                pending.append(pool.submit(stage, i, *w))
                if len(pending) >= depth:
                    break
            while pending:
                b, ev = pending.pop(0).result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(stage, nxt[0], *nxt[1]))
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                slot = getattr(b, "_slot", None)
                if slot is None:                         # (per-batch device tensors: keep the allocator informed)
                    for v in b.__dict__.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(cur)
                free.put((b._staging, ev))
                self._pinned = [(t, e) for t, e in self._pinned if t is not b._staging] + [(b._staging, ev)]
                yield b
                if slot is not None:                     # the consumer has enqueued its use of the landing buffer
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                    consumed[slot] = done


def write_shards(directory: str, batches: Iterable[GraphBatch], prefix: str = "shard") -> List[str]:
    """One shard per collated batch of ``batches`` (e.g. a few hundred thousand molecules each); returns the paths."""
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i, b in enumerate(batches):
        path = os.path.join(directory, f"{prefix}-{i:05d}.mkgs")
        write_shard(path, b)
        paths.append(path)
    return paths
