"""Seeded synthetic molecules shaped like the PubChem assays the reference trains on.

No assay data exists on the build or GPU boxes (reference ``README.md:29``: the
SDF files are a separate download), so every measurement runs on synthetic
molecules with the shape statistics fixed in SURVEY.md section 8(d):

* atoms per molecule ``clip(round(N(25, 6)), 8, 60)``;
* a random spanning tree plus 1-3 ring-closure bonds, maximum degree 4,
  minimum degree 1, degree mix close to 22/44/30/4 % for degrees 1..4;
* each bond stored as two consecutive directed edges with identical attributes
  (reference ``wrapper.py:152-156``);
* ``x ~ N(0,1)^28`` (stand-in for batch-normalised atom features),
  ``edge_attr`` = one-hot(4 bond types) + 3 Bernoulli flags (``wrapper.py:139-150``),
  ``p ~ N(0, 1.5^2)^3``, ``y ~ Bernoulli(active fraction)``.

Assay sizes (reference ``utils/data_split.py:68-79``).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .receptive_field import GraphBatch, attach_receptive_fields

# molecules per assay: actives + inactives (reference utils/data_split.py:68-79)
# (actives, inactives) of the nine PubChem assays of the benchmark set (wrapper.py:351-360) and the toy set 9999
ASSAY_COUNTS = {
    "435008": (233, 217923), "1798": (187, 61645), "435034": (362, 61393), "1843": (172, 301318), "2258": (213, 302189),
    "463087": (703, 100171), "488997": (252, 302051), "2689": (172, 319617), "485290": (278, 341026), "9999": (37, 226),
}
NINE_ASSAYS = ("435008", "1798", "435034", "1843", "2258", "463087", "488997", "2689", "485290")
ASSAY_SIZES = {k: a + i for k, (a, i) in ASSAY_COUNTS.items()}
ASSAY_SIZES["all9"] = sum(ASSAY_SIZES[k] for k in NINE_ASSAYS)          # 2 009 905 molecules (BASELINE configs[4])
ASSAY_ACTIVE_FRACTION = {k: a / float(a + i) for k, (a, i) in ASSAY_COUNTS.items()}
ASSAY_ACTIVE_FRACTION["9999"] = 0.1

NODE_DIM = 28
EDGE_DIM = 7
MAX_ATOMS = 60
MIN_ATOMS = 8
P_CHAIN = 0.35   # growth: probability of extending the chain at its tip
RINGS_LO, RINGS_HI = 2, 4   # ring-closure bonds per molecule (uniform)
P_RING4 = 0.50   # ring closure: probability that it may land on a degree-3 atom
P_CAP4 = 0.27    # growth: probability that a branch may create a degree-4 atom


def _random_graphs(rng: np.random.Generator, n_atoms: np.ndarray):
    """Vectorised (over molecules) bounded-degree tree growth + ring closures.

    Returns a list of bond arrays as (mol_id, i, j) triples in creation order.
    """
    b = n_atoms.shape[0]
    deg = np.zeros((b, MAX_ATOMS), dtype=np.int64)
    rows = np.arange(b)
    bonds_mol, bonds_i, bonds_j = [], [], []
    # growth: atom t attaches to an earlier atom; mostly the previous one
    # (chains/rings dominate drug-like graphs), otherwise a random earlier atom
    # that still has room; branching atoms are capped at 3 so that degree 4
    # stays rare.
    for t in range(1, MAX_ATOMS):
        live = n_atoms > t
        if not live.any():
            break
        idx = rows[live]
        k = idx.shape[0]
        d_prev = deg[idx, t - 1]
        # random earlier atom that still has room under a per-draw cap
        cap = np.where(rng.random((k, 1)) < P_CAP4, 4, 3)
        score = np.where(deg[idx, :t] < cap, rng.random((k, t)), -1.0)
        cand = score.argmax(axis=1)
        full = score.max(axis=1) < 0
        if full.any():
            # a tree always has an atom of degree < 4
            score2 = np.where(deg[idx, :t] < 4, rng.random((k, t)), -1.0)
            cand = np.where(full, score2.argmax(axis=1), cand)
        use_prev = (rng.random(k) < P_CHAIN) & (d_prev < 2)
        parent = np.where(use_prev | (t == 1), t - 1, cand)
        deg[idx, parent] += 1
        deg[idx, t] += 1
        bonds_mol.append(idx)
        bonds_i.append(parent)
        bonds_j.append(np.full(k, t))
    # ring closures: 1-3 per molecule between atoms 4-6 apart in growth order
    n_rings = rng.integers(RINGS_LO, RINGS_HI + 1, size=b)
    adj = np.zeros((b, MAX_ATOMS, MAX_ATOMS), dtype=bool) if b <= 8192 else None
    if adj is not None:
        for m_, i_, j_ in zip(bonds_mol, bonds_i, bonds_j):
            adj[m_, i_, j_] = True
            adj[m_, j_, i_] = True
    for r in range(RINGS_HI):
        want = n_rings > r
        idx = rows[want]
        if idx.size == 0:
            continue
        k = idx.shape[0]
        span = rng.integers(4, 7, size=k)
        # close the ring at a leaf when there is one (turns a degree-1 atom
        # into a ring atom), onto an atom `span` steps away in growth order
        col = np.arange(MAX_ATOMS)[None, :]
        valid = col < n_atoms[idx, None]
        leaf_score = np.where(valid & (deg[idx] == 1), rng.random((k, MAX_ATOMS)), -1.0)
        any_score = np.where(valid & (deg[idx] < 3), rng.random((k, MAX_ATOMS)), -1.0)
        i = np.where(leaf_score.max(axis=1) >= 0, leaf_score.argmax(axis=1), any_score.argmax(axis=1))
        j = np.where(i >= span, i - span, np.minimum(i + span, n_atoms[idx] - 1))
        capj = np.where(rng.random(k) < P_RING4, 4, 3)
        ok = (i != j) & (deg[idx, i] < 3) & (deg[idx, j] < capj)
        if adj is not None:
            ok &= ~adj[idx, i, j]
        idx, i, j = idx[ok], i[ok], j[ok]
        deg[idx, i] += 1
        deg[idx, j] += 1
        if adj is not None:
            adj[idx, i, j] = True
            adj[idx, j, i] = True
        bonds_mol.append(idx)
        bonds_i.append(i)
        bonds_j.append(j)
    return (np.concatenate(bonds_mol), np.concatenate(bonds_i), np.concatenate(bonds_j))


def make_batch(num_molecules: int, seed: int, *, assay: str = "1798",
               duplicate_fraction: float = 0.0, device: Optional[torch.device] = None,
               with_receptive_fields: bool = True) -> GraphBatch:
    """One collated batch of ``num_molecules`` synthetic molecules.

    ``duplicate_fraction`` > 0 copies the feature row of one neighbour onto a
    sibling for that fraction of atoms, which creates the permutation ties the
    reference meets on symmetric substituents.
    """
    rng = np.random.default_rng(seed)
    n_atoms = np.clip(np.rint(rng.normal(25.0, 6.0, size=num_molecules)), MIN_ATOMS, MAX_ATOMS).astype(np.int64)
    chunks = []
    # the dense adjacency used to reject duplicate ring bonds is O(B*60*60);
    # grow big batches in slices
    step = 8192
    for s in range(0, num_molecules, step):
        bm, bi, bj = _random_graphs(rng, n_atoms[s:s + step])
        chunks.append((bm + s, bi, bj))
    bm = np.concatenate([c[0] for c in chunks])
    bi = np.concatenate([c[1] for c in chunks])
    bj = np.concatenate([c[2] for c in chunks])
    # group bonds by molecule, creation order kept inside a molecule
    order = np.argsort(bm, kind="stable")
    bm, bi, bj = bm[order], bi[order], bj[order]
    offset = np.zeros(num_molecules + 1, dtype=np.int64)
    offset[1:] = np.cumsum(n_atoms)
    n = int(offset[-1])
    gi = bi + offset[bm]
    gj = bj + offset[bm]
    nb = gi.shape[0]
    edge_index = np.empty((2, 2 * nb), dtype=np.int64)
    edge_index[0, 0::2] = gi
    edge_index[1, 0::2] = gj
    edge_index[0, 1::2] = gj
    edge_index[1, 1::2] = gi
    bond_type = rng.choice(4, size=nb, p=[0.62, 0.25, 0.11, 0.02])
    bond_attr = np.zeros((nb, EDGE_DIM), dtype=np.float32)
    bond_attr[np.arange(nb), bond_type] = 1.0
    bond_attr[:, 4] = (bond_type == 1)
    bond_attr[:, 5] = rng.random(nb) < 0.45
    bond_attr[:, 6] = rng.random(nb) < 0.55
    edge_attr = np.repeat(bond_attr, 2, axis=0)
    x = rng.standard_normal((n, NODE_DIM)).astype(np.float32)
    if duplicate_fraction > 0.0:
        # make pairs of neighbours of one atom carry identical feature rows
        src = edge_index[0]
        order_e = np.argsort(src, kind="stable")
        degs = np.bincount(src, minlength=n)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(degs)
        cand = np.nonzero(degs >= 2)[0]
        pick = cand[rng.random(cand.shape[0]) < duplicate_fraction]
        a = edge_index[1, order_e[rowptr[pick]]]
        c = edge_index[1, order_e[rowptr[pick] + 1]]
        x[c] = x[a]
    p = (1.5 * rng.standard_normal((n, 3))).astype(np.float32)
    if assay == "all9":
        # a mixed batch of the nine assays concatenated (BASELINE configs[4]): every molecule drawn from one of them in
        # proportion to its size, labelled at that assay's active rate.  (The synthetic molecules themselves have one
        # distribution whatever the assay: there is no per-assay chemistry offline.)
        sizes = np.array([ASSAY_SIZES[k] for k in NINE_ASSAYS], dtype=np.float64)
        which = rng.choice(len(NINE_ASSAYS), size=num_molecules, p=sizes / sizes.sum())
        rate = np.array([ASSAY_ACTIVE_FRACTION[k] for k in NINE_ASSAYS])[which]
        assay_id = np.array([int(k) for k in NINE_ASSAYS], dtype=np.int64)[which]
    else:
        rate = ASSAY_ACTIVE_FRACTION.get(assay, 0.003)
        assay_id = np.full(num_molecules, int(assay) if str(assay).isdigit() else 0, dtype=np.int64)
    y = (rng.random(num_molecules) < rate).astype(np.float32)
    batch_vec = np.repeat(np.arange(num_molecules, dtype=np.int64), n_atoms)
    out = GraphBatch(
        x=torch.from_numpy(x), p=torch.from_numpy(p),
        edge_index=torch.from_numpy(edge_index), edge_attr=torch.from_numpy(edge_attr),
        batch=torch.from_numpy(batch_vec), y=torch.from_numpy(y),
        num_graphs=num_molecules, smiles=None, assay_id=torch.from_numpy(assay_id))
    if device is not None:
        out = out.to(device)
    if with_receptive_fields:
        attach_receptive_fields(out)
    return out


def degree_histogram(batch: GraphBatch):
    deg = torch.bincount(batch.edge_index[0], minlength=batch.x.shape[0])
    return torch.bincount(deg, minlength=6).tolist()
