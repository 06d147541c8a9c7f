"""Golden vectors for the receptive-field builder from the reference itself.

Runs ONLY in the build container (the reference checkout is at /root/reference; nothing in tests/, smoke() or
bench.py reads it).  ``/root/reference/wrapper.py`` is imported unmodified; the packages it imports at module level that
are absent here are replaced by empty stand-ins first (they are not touched by the class exercised):

* ``rdkit`` (+ the sub-modules named in wrapper.py:5-17), ``tqdm``, ``models.ChIRoNet.embedding_functions``: never
  called by ``ToXAndPAndEdgeAttrForDeg``;
* ``torch_geometric.data.{InMemoryDataset, Data}``, ``torch_geometric.data.collate.collate``: attribute bag / unused;
* ``torch_geometric.utils.degree``: PyG's documented behaviour -- the number of occurrences of every node id in
  ``index`` as a float tensor of length ``num_nodes`` (wrapper.py:574-576 is its only use).

The reference runs the transform per molecule (``pre_transform``) and PyG's collation then concatenates the
per-molecule tensors, adding the molecule's node offset to every attribute whose name contains ``index`` (PyG
``Data.__inc__``) and concatenating ``edge_index`` along its last dimension (``__cat_dim__``).  That collation rule is
restated here (PyG itself is absent: "parity unpinned" at the PyG boundary) and applied to the REFERENCE's
per-molecule outputs; both the per-molecule outputs and the collated batch are stored.

Usage:  python tests/golden/make_golden_rf.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)


class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def install_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    rd = mod("rdkit")
    chem = mod("rdkit.Chem")
    rd.Chem = chem
    rd.RDLogger = mod("rdkit.RDLogger")
    for sub in ("AllChem", "EState", "rdMolDescriptors", "rdPartialCharges"):
        setattr(chem, sub, mod("rdkit.Chem." + sub))
    mod("tqdm", tqdm=lambda it, *a, **k: it)
    mod("models")
    mod("models.ChIRoNet")
    mod("models.ChIRoNet.embedding_functions", embedConformerWithAllPaths=None)
    tg = mod("torch_geometric")
    tg.data = mod("torch_geometric.data", InMemoryDataset=object, Data=Data)
    mod("torch_geometric.data.collate", collate=None)

    def degree(index, num_nodes=None, dtype=None):
        n = int(index.max()) + 1 if num_nodes is None else num_nodes
        out = torch.zeros(n, dtype=dtype or torch.float)
        return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))
    tg.utils = mod("torch_geometric.utils", degree=degree)


def import_wrapper():
    install_stand_ins()
    spec = importlib.util.spec_from_file_location("reference_wrapper", os.path.join(REFERENCE, "wrapper.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


FIELDS = [f"{nm}_deg{d}" for d in range(1, 5) for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index")]


def molecule(seed, n_atoms, extra_bonds, hub=False, shuffle=False):
    """A connected synthetic molecule: random tree + ring closures, bonds stored as consecutive (i, j), (j, i)."""
    g = torch.Generator().manual_seed(seed)
    pairs = []
    deg = [0] * n_atoms
    for v in range(1, n_atoms):
        while True:
            u = 0 if (hub and v <= 7) else int(torch.randint(0, v, (1,), generator=g))     # hub: atom 0 gets 7 bonds (in no bucket)
            if (hub and u == 0 and v <= 7) or deg[u] < 4:
                break
        pairs.append((u, v)); deg[u] += 1; deg[v] += 1
    tries = 0
    while extra_bonds and tries < 200:
        tries += 1
        u, v = [int(t) for t in torch.randint(0, n_atoms, (2,), generator=g)]
        if u != v and deg[u] < 4 and deg[v] < 4 and (u, v) not in pairs and (v, u) not in pairs:
            pairs.append((u, v)); deg[u] += 1; deg[v] += 1; extra_bonds -= 1
    if shuffle:
        perm = torch.randperm(len(pairs), generator=g).tolist()
        pairs = [pairs[i] for i in perm]
    ei = torch.tensor([[a, b] for (a, b) in pairs for (a, b) in ((a, b), (b, a))], dtype=torch.long).t().contiguous()
    ea = torch.rand(len(pairs), 7, generator=g).repeat_interleave(2, dim=0)
    return Data(x=torch.randn(n_atoms, 28, generator=g), p=torch.randn(n_atoms, 3, generator=g), edge_index=ei, edge_attr=ea)


def collate(mols):
    """PyG collation of the reference's per-molecule objects (see the module docstring)."""
    out = {}
    offs = np.cumsum([0] + [m.x.shape[0] for m in mols])
    for k in ["x", "p", "edge_attr"] + FIELDS:
        parts = []
        for m, o in zip(mols, offs):
            v = getattr(m, k)
            if v.numel() == 0:
                continue
            parts.append(v + int(o) if "index" in k else v)
        if parts:
            out[k] = torch.cat(parts, dim=0)
        else:
            out[k] = torch.zeros(0, dtype=torch.long) if "index" in k else torch.zeros(0)
    out["edge_index"] = torch.cat([m.edge_index + int(o) for m, o in zip(mols, offs)], dim=1)
    out["batch"] = torch.cat([torch.full((m.x.shape[0],), i, dtype=torch.long) for i, m in enumerate(mols)])
    return out


def main():
    w = import_wrapper()
    transform = w.ToXAndPAndEdgeAttrForDeg()
    mols = [molecule(101, 23, 2), molecule(102, 9, 0), molecule(103, 31, 3, shuffle=True), molecule(104, 14, 1),
            molecule(105, 12, 0, hub=True), molecule(106, 2, 0), molecule(107, 40, 3, shuffle=True)]
    arrays = {}
    for i, m in enumerate(mols):
        for k in ("x", "p", "edge_index", "edge_attr"):
            arrays[f"mol{i}/in_{k}"] = getattr(m, k).numpy()
        transform(m)                                  # the reference's own per-molecule transform (wrapper.py:637-672)
        for k in FIELDS:
            arrays[f"mol{i}/{k}"] = getattr(m, k).numpy()
    batch = collate(mols)
    for k, v in batch.items():
        arrays[f"batch/{k}"] = v.numpy()
    arrays["num_molecules"] = np.array(len(mols))
    path = os.path.join(HERE, "g10_receptive_fields.npz")
    np.savez_compressed(path, **arrays)
    degs = torch.bincount(batch["edge_index"][0], minlength=batch["x"].shape[0])
    print(f"wrote g10_receptive_fields.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays; "
          f"{batch['x'].shape[0]} atoms, degree histogram {torch.bincount(degs).tolist()}")


if __name__ == "__main__":
    main()
