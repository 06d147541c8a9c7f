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

Layout (version 1)::

    header   256 bytes: b"MKGS", u32 version, u64 n_molecules, u64 n_atoms, u64 n_bonds (directed edges),
                        u32 x_dim, u32 e_dim, u32 p_dim, u32 reserved, then 9 x u64 array offsets in the order below
    mol_atom_ptr  i64 [M + 1]     first atom of every molecule
    mol_edge_ptr  i64 [M + 1]     first directed edge of every molecule
    y             f32 [M]         label (data.py:37 trains on one task)
    assay_id      i32 [M]         PubChem AID the molecule comes from (mixed nine-assay shards, BASELINE configs[4])
    x             f32 [A, x_dim]
    p             f32 [A, p_dim]
    edge_src      i32 [E]         shard-global atom ids; bonds as consecutive (i, j), (j, i) (wrapper.py:152-156)
    edge_dst      i32 [E]
    edge_attr     f32 [E, e_dim]
"""
from __future__ import annotations

import os
import struct
from queue import Queue
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .receptive_field import GraphBatch

MAGIC = b"MKGS"
VERSION = 1
HEADER_BYTES = 256
ALIGN = 64
_ARRAYS = ("mol_atom_ptr", "mol_edge_ptr", "y", "assay_id", "x", "p", "edge_src", "edge_dst", "edge_attr")
_HEAD = struct.Struct("<4sIQQQIIII9Q")


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
    arrays = {"mol_atom_ptr": atom_ptr, "mol_edge_ptr": edge_ptr, "y": y, "assay_id": assay,
              "x": np.ascontiguousarray(x), "p": np.ascontiguousarray(p),
              "edge_src": ei[0].astype(np.int32), "edge_dst": ei[1].astype(np.int32), "edge_attr": np.ascontiguousarray(ea)}
    offsets, off = [], HEADER_BYTES
    for name in _ARRAYS:
        offsets.append(off)
        off = _align(off + arrays[name].nbytes)
    head = _HEAD.pack(MAGIC, VERSION, m, a, e, x.shape[1], ea.shape[1] if ea.ndim == 2 else 0, p.shape[1], 0, *offsets)
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
        magic, version, m, a, e, xd, ed, pd, _, *offs = _HEAD.unpack(head[:_HEAD.size])
        if magic != MAGIC:
            raise ValueError(f"{path}: not a molecule shard (magic {magic!r})")
        if version != VERSION:
            raise ValueError(f"{path}: shard version {version}, this reader knows {VERSION}")
        self.n_molecules, self.n_atoms, self.n_edges = int(m), int(a), int(e)
        self.x_dim, self.e_dim, self.p_dim = int(xd), int(ed), int(pd)
        shapes = {"mol_atom_ptr": (np.int64, (m + 1,)), "mol_edge_ptr": (np.int64, (m + 1,)), "y": (np.float32, (m,)),
                  "assay_id": (np.int32, (m,)), "x": (np.float32, (a, xd)), "p": (np.float32, (a, pd)),
                  "edge_src": (np.int32, (e,)), "edge_dst": (np.int32, (e,)), "edge_attr": (np.float32, (e, ed))}
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


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class ShardLoader:
    """Batches of ``batch_size`` consecutive molecules from a sequence of shards, for rank ``rank`` of ``world``.

    Batch k of the global sequence (shard after shard, a shard's tail batch may be short) goes to rank ``k % world``
    (SURVEY 8e).  ``workers`` background threads stage the next batches into pinned buffers and issue their
    host-to-device copies on a copy stream, ``prefetch`` batches ahead, handed over in order; the consumer's stream waits
    for the copy's event, not for the host.
    """

    def __init__(self, paths: Sequence[str], batch_size: int, device="cpu", rank: int = 0, world: int = 1,
                 prefetch: int = 2, drop_last: bool = False, workers: int = 2):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} of world {world}")
        self.paths, self.batch_size, self.device = list(paths), int(batch_size), torch.device(device)
        self.rank, self.world, self.prefetch, self.drop_last = rank, world, max(1, int(prefetch)), drop_last
        self.workers = max(1, int(workers))
        self.shards = [Shard(p) for p in self.paths]

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
                yield collate(self.shards[si], m0, m1, self.device)
            return
        from concurrent.futures import ThreadPoolExecutor
        copy_stream = torch.cuda.Stream(device=self.device)
        depth = self.prefetch + self.workers
        free: Queue = Queue()                            # pinned staging buffers with the event of their last copy
        for _ in range(depth + 1):
            free.put((None, None))

        def stage(si, m0, m1):
            staging, last = free.get()
            if last is not None:
                last.synchronize()                       # the buffer's previous copy has left the host
            with torch.cuda.device(self.device):
                b = collate(self.shards[si], m0, m1, self.device, staging, copy_stream)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return b, ev

        with ThreadPoolExecutor(self.workers) as pool:   # (numpy's slice copies release the GIL: the workers overlap)
            pending = []
            it = iter(work)
            for w in it:
                pending.append(pool.submit(stage, *w))
                if len(pending) >= depth:
                    break
            while pending:
                b, ev = pending.pop(0).result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(stage, *nxt))
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for v in b.__dict__.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
                free.put((b._staging, ev))
                yield b


def write_shards(directory: str, batches: Iterable[GraphBatch], prefix: str = "shard") -> List[str]:
    """One shard per collated batch of ``batches`` (e.g. a few hundred thousand molecules each); returns the paths."""
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i, b in enumerate(batches):
        path = os.path.join(directory, f"{prefix}-{i:05d}.mkgs")
        write_shard(path, b)
        paths.append(path)
    return paths
