"""The molecule-resident small-batch path (``include/molkgnn_hip.h``: ``mkgnn_molecule_step``; ``csrc/kgnn_molecule.hip``).

At the reference's own batch sizes (``README.md:81`` ``--batch_size 16``; BASELINE configs[0] / [2]) the step of one launch
per operator is ~30 dependent launches of a few microseconds each.  Here ``MolKGNNNet.forward`` -- batch norm, every
``KernelSetConv`` + ``propagate``, readout (reference ``MolKGNNNet.py:115-146``, ``KernelLayer.py:109-120``) -- is ONE launch
in which a workgroup owns a chunk of whole molecules, and ``GNNModel.loss`` (``model.py:150, 169, 190-198``) is one launch
for forward + loss + backward; a preparation launch in front (unit kernel rows, partial batch-norm statistics) and a
reduction launch behind (fixed-order sums of the workgroups' partial gradients).

This module holds the host side: the per-batch chunk table (``MoleculePlan``: which molecules a workgroup takes; checked
once per batch that the bonds are stored in both directions, molecules are contiguous and small enough), the ctypes
description of the network, and the two autograd operators.  No CPU path: everything raises without the library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from . import _lib
from .plan import BatchPlan

MAX_ATOMS = _lib.MOLECULE_MAX_ATOMS
MAX_MOLS = _lib.MOLECULE_MAX_MOLS
# MKGNN_MOLECULE: unset -- the one-launch TRAINING step where it is the faster one: inside a hipGraph capture up to
# MKGNN_MOLECULE_AUTO_MOLS (32) molecules, launched eagerly up to MKGNN_MOLECULE_MAX_MOLS (512) on batches that come back
# (a batch seen for the first time above 32 molecules takes the per-operator kernels: _ready); '1' -- whenever the model and
# the batch qualify, up to MKGNN_MOLECULE_MAX_MOLS, and MolKGNNNet.forward alone as well; '0' -- never.  Measured on MI355X
# (round 4).  Replayed graphs: 0.23-0.25 ms per step at 16 molecules against 0.255 per operator, but 0.33 against
# 0.289 ms at 256 -- a workgroup per molecule is bound by its own fp32 matrix work, and with one chunk per CU the launch lasts
# as long as its slowest chunk, where the per-operator kernels spread a layer over the whole chip.  Eager steps (the
# reference's own training-loop style, host-bound): 1.4 ms against 1.95 ms per operator from 64 to 512 molecules -- 4 launches
# and one autograd node instead of ~36 and a dozen.
_MODE = os.environ.get("MKGNN_MOLECULE", "")
_MAX_MOLS_AUTO = int(os.environ.get("MKGNN_MOLECULE_AUTO_MOLS", "32"))
_MAX_MOLS_FORCED = int(os.environ.get("MKGNN_MOLECULE_MAX_MOLS", "512"))
debug_capture: Optional[dict] = None      # tests: a dict here receives the pair records / sim rows of the next forward


class MoleculePlan:
    """Chunk table of one batch: consecutive whole molecules, at most ``cap`` atoms (and 16 molecules) per chunk."""

    def __init__(self, chunk_ptr, mol_ptr, atom_deg, atom_rank, n_mols, n_chunks, max_chunk_atoms):
        self.chunk_ptr, self.mol_ptr, self.atom_deg, self.atom_rank = chunk_ptr, mol_ptr, atom_deg, atom_rank
        self.n_mols, self.n_chunks, self.max_chunk_atoms = n_mols, n_chunks, max_chunk_atoms


def chunk_molecules(sizes, cap: int):
    """Greedy chunk table: consecutive whole molecules while the chunk stays within ``cap`` atoms (a molecule larger than
    ``cap`` gets a chunk of its own) and ``MAX_MOLS`` molecules.  Returns (chunk_ptr: first molecule of every chunk + the
    molecule count, atoms of the largest chunk)."""
    chunk_ptr, cur_atoms, cur_mols, max_atoms = [0], 0, 0, 0
    for g, sz in enumerate(sizes):
        if cur_mols and (cur_atoms + sz > max(cap, sz) or cur_mols >= MAX_MOLS):
            chunk_ptr.append(g)
            max_atoms = max(max_atoms, cur_atoms)
            cur_atoms, cur_mols = 0, 0
        cur_atoms += sz
        cur_mols += 1
    chunk_ptr.append(len(sizes))
    return chunk_ptr, max(max_atoms, cur_atoms)


def build_molecule_plan(plan: BatchPlan, batch_vec: torch.Tensor, n_mols: Optional[int], cap: Optional[int] = None) -> Optional[MoleculePlan]:
    """The chunk table, or ``None`` when the batch does not qualify.  One host synchronisation (the validity flag and the
    molecule sizes in one transfer); call it outside captures -- ``molecule_plan`` caches the result on the batch's index
    plan."""
    dev = plan.device
    n = plan.n_atoms
    if dev.type != "cuda" or n < 1 or batch_vec is None or batch_vec.numel() != n or plan.edge_index is None:
        return None
    for b in plan.buckets:                               # (host-side facts first: no device work for a batch that cannot qualify)
        if b.count and b.e_unit(8 if b.e_nei is None else int(b.e_nei.shape[-1])) is None:
            return None
    ei = plan.edge_index
    bv = batch_vec.long()
    if n_mols is None:
        n_mols = int(bv.max().item()) + 1               # (a loader that knows the molecule count passes it: data.num_graphs)
    n_mols = int(n_mols)
    src, dst = ei[0].long(), ei[1].long()
    deg = torch.bincount(src, minlength=n)
    true = torch.ones((), dtype=torch.bool, device=dev)
    # every check is left on the device and read back in ONE transfer together with the molecule sizes (round 4 read each
    # of them back by itself: eight host round trips per fresh batch)
    checks = [(bv[1:] >= bv[:-1]).all() if n > 1 else true, (deg <= 4).all()]
    if src.numel():
        keys, rkeys = src * n + dst, dst * n + src
        sk = torch.sort(keys).values
        checks.append((bv[src] == bv[dst]).all())
        checks.append((sk == torch.sort(rkeys).values).all())                 # every bond stored in both directions
        checks.append((sk[1:] != sk[:-1]).all() if sk.numel() > 1 else true)  # ... and only once
    # the buckets must be what the edge list says (sum of the buckets' atoms = atoms of degree 1..4)
    checks.append((deg >= 1).sum() == plan.n_focal)
    checks.append((bv.max() < n_mols) & (bv.min() >= 0))
    ok = torch.stack(checks).all().to(torch.int64).reshape(1)
    host = torch.cat([ok, torch.bincount(bv.clamp(0, n_mols - 1), minlength=n_mols)]).tolist()
    if not host[0]:
        return None
    sizes = host[1:]
    if len(sizes) != n_mols or max(sizes) > MAX_ATOMS:
        return None
    if cap is None:
        cap = 32 if n_mols <= 512 else MAX_ATOMS
    chunk_ptr, max_atoms = chunk_molecules(sizes, cap)
    mol_ptr = torch.zeros(n_mols + 1, dtype=torch.int64)
    mol_ptr[1:] = torch.tensor(sizes, dtype=torch.int64).cumsum(0)
    atom_deg = torch.zeros(n, dtype=torch.int8, device=dev)
    atom_rank = torch.zeros(n, dtype=torch.int32, device=dev)
    for b in plan.buckets:
        if b.count:
            atom_deg[b.sel] = b.degree
            atom_rank[b.sel] = torch.arange(b.count, dtype=torch.int32, device=dev)
    return MoleculePlan(torch.tensor(chunk_ptr, dtype=torch.int32).to(dev), mol_ptr.to(dev), atom_deg, atom_rank,
                        n_mols, len(chunk_ptr) - 1, max(max_atoms, 1))


def molecule_plan(plan: BatchPlan, batch_vec, n_mols) -> Optional[MoleculePlan]:
    cached = getattr(plan, "_molecule", None)
    if cached is None:
        if torch.cuda.is_current_stream_capturing():
            return None                                  # (first seen inside a capture: the per-operator path; no host round trip here)
        mp = build_molecule_plan(plan, batch_vec, n_mols)
        cached = plan._molecule = (mp,)
    return cached[0]


# ------------------------------------------------------------------------------------------------ the model side ----
def _layer_params(layer) -> Optional[List[torch.Tensor]]:
    if any(k is not None for k in layer.fixed_kernelconv_set):
        return None
    return layer._bank_params("train", next(iter(layer.parameters())))[0]


_PARAM_CACHE_ATTR = "_mkgnn_molecule_params"


def flat_parameters(net, ffn=None):
    """The operator's parameters in a fixed order: 7 tensors per degree per layer, batch norm weight / bias, lin1, lin2
    (weight, bias each), then the head's weight / bias.  ``None`` for an absent optional tensor.  Cached on the module (an
    eager step at batch 16 is host-bound: rebuilding the list twice per step was a quarter of it); the cache is keyed by the
    identity of every parameter, so swapping a parameter or a kernel set rebuilds it."""
    cached = getattr(net, _PARAM_CACHE_ATTR, None)
    key = (id(ffn),) + tuple(id(p) for p in net.parameters()) + (() if ffn is None else tuple(id(p) for p in ffn.parameters()))
    if cached is not None and cached[0] == key:
        return cached[1]
    out = _flat_parameters(net, ffn)
    try:
        object.__setattr__(net, _PARAM_CACHE_ATTR, (key, out))
    except Exception:
        pass
    return out


def _flat_parameters(net, ffn=None):
    out = []
    for layer in net.gnn.layers:
        lp = _layer_params(layer)
        if lp is None:
            return None
        out += lp
    bn = net.node_batch_norm
    out += [bn.weight, bn.bias, net.graph_embedding_lin1.weight, net.graph_embedding_lin1.bias,
            net.graph_embedding_lin2.weight, net.graph_embedding_lin2.bias]
    if ffn is not None:
        out += [ffn.weight, ffn.bias]
    return out


def _f32(t):
    return t is None or (t.dtype == torch.float32 and t.is_cuda and t.is_contiguous())


def model_qualifies(net, data, ffn=None) -> bool:
    """Does the model's shape (and state) fit the molecule-resident kernels?"""
    if getattr(data, 'n_valid_atoms', None) is not None or getattr(data, 'n_valid_molecules', None) is not None:
        return False                                     # fixed-shape (padded) batches: the per-operator path
    x = data.x
    if not x.is_cuda or x.dtype != torch.float32 or x.requires_grad or x.dim() != 2 or x.stride(1) != 1:
        return False
    if len(net.gnn.layers) > _lib.MOLECULE_MAX_LAYERS:
        return False
    if net.dropout.training and net.dropout.p > 0.0:
        return False                                     # readout dropout: not in these kernels (the reference's default is 0)
    bn = net.node_batch_norm
    if bn.momentum is None or (not bn.training and bn.running_mean is None):
        return False
    params = flat_parameters(net, ffn)
    if params is None:
        return False
    # dtype / device / layout of ~90 tensors, every call: nn.Module.to() / .half() / .double() change p.data in place and keep
    # the Parameter objects, so nothing keyed by their identity may remember the answer
    xdev = x.device
    for p in params:
        if p is not None and (p.dtype != torch.float32 or p.device != xdev or not p.is_contiguous()):
            return False
    if ffn is not None and (ffn.out_features != 1):
        return False
    key = (tuple(tuple(layer.L) for layer in net.gnn.layers), x.shape[1], tuple(net.graph_embedding_lin1.weight.shape),
           tuple(net.graph_embedding_lin2.weight.shape), int(params[2].shape[-1]) if params[2] is not None else 0,
           None if ffn is None else (ffn.in_features, ffn.out_features, ffn.bias is not None))
    ok = _SHAPE_OK.get(key)
    if ok is None:
        st, keep = _net_struct(net, ffn, params, None, None, None, 0.0, True)
        ok = _SHAPE_OK[key] = bool(_lib.load().mkgnn_molecule_supported(C.byref(st), int(x.shape[1])))
        del keep
    return ok


_SHAPE_OK: dict = {}


def wanted(n_mols: int) -> bool:
    if _MODE == "0":
        return False
    if _MODE == "1" or not torch.cuda.is_current_stream_capturing():
        return n_mols <= _MAX_MOLS_FORCED
    return n_mols <= _MAX_MOLS_AUTO


_EDGE_STATS_ROWS = 8192          # bond rows the preparation launch's extra block takes (kgnn_molecule.hip MOL_EDGE_STATS_ROWS)


def _edge_stats_in_launch(es) -> bool:
    """Whether ``edge_batch_norm``'s statistics (``MolKGNNNet._edge_stats``) ride in the step's preparation launch."""
    if es is None:
        return False
    ea, bn = es[0], es[1]
    return (ea.is_cuda and ea.dim() == 2 and 1 <= ea.shape[0] <= _EDGE_STATS_ROWS and 1 <= ea.shape[1] <= 8
            and bn.momentum is not None and bn.running_mean is not None and bn.running_mean.dtype == torch.float32)


def _net_struct(net, ffn, params, grads, saved, sims, head_dropout, update_running, rng=None, rng_used=None, edge_stats=None):
    """``mkgnn_molecule_net`` for the model; ``grads`` / ``saved`` / ``sims``: lists parallel to the layers (or None);
    ``edge_stats``: ``MolKGNNNet._edge_stats(data)`` where ``_edge_stats_in_launch`` holds."""
    st = _lib.MoleculeNet()
    layers = net.gnn.layers
    st.num_layers = len(layers)
    keep = [params]
    k = 0
    E = 1
    for li, layer in enumerate(layers):
        Y = st.layer[li] if li < _lib.MOLECULE_MAX_LAYERS else None
        lp = params[k:k + 28]
        k += 28
        if Y is None:
            continue
        E = int(lp[2].shape[-1])
        Y.F = int(lp[0].shape[1])
        for d in range(4):
            xc, xs, es, ps, w_s, w_c, w_e = lp[7 * d:7 * d + 7]
            b = Y.bank[d]
            b.num_kernels = int(xc.shape[0])
            b.x_center, b.x_support, b.edge_attr_support = _lib.ptr(xc), _lib.ptr(xs), _lib.ptr(es)
            b.p_support = _lib.ptr(ps)
            b.support_attr_sc_weight, b.center_attr_sc_weight, b.edge_attr_support_sc_weight = w_s.data_ptr(), w_c.data_ptr(), w_e.data_ptr()
            if grads is not None and grads[li][d] is not None:
                g = Y.grad[d]
                gx, gs, ge, gth = grads[li][d]
                g.x_center, g.x_support, g.edge_attr_support = gx.data_ptr(), gs.data_ptr(), ge.data_ptr()
                g.support_attr_sc_weight, g.center_attr_sc_weight, g.edge_attr_support_sc_weight = \
                    gth.data_ptr(), gth.data_ptr() + 4, gth.data_ptr() + 8
            if saved is not None and saved[li][d] is not None:
                pr, ch = saved[li][d]
                Y.saved[d].pair_state = _lib.ptr(pr)
                Y.saved[d].chirality = _lib.ptr(ch)
        if sims is not None and sims[li] is not None:
            Y.sim_out, Y.sim_stride = sims[li].data_ptr(), int(sims[li].stride(0))
    st.E = E
    bn = net.node_batch_norm
    bw, bb, w1, b1, w2, b2 = params[k:k + 6]
    st.bn_weight, st.bn_bias = _lib.ptr(bw), _lib.ptr(bb)
    use_batch = bn.training or bn.running_mean is None
    st.bn_training = 1 if use_batch else 0
    if bn.running_mean is not None and (update_running and bn.training or not use_batch):
        st.bn_running_mean, st.bn_running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        if update_running and bn.training and bn.num_batches_tracked is not None:
            st.bn_num_batches_tracked = bn.num_batches_tracked.data_ptr()
    st.bn_eps, st.bn_momentum = float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.0)
    ro = st.readout
    ro.lin1_weight, ro.lin1_bias, ro.lin2_weight, ro.lin2_bias = _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2)
    ro.H, ro.F = int(w1.shape[0]), int(w1.shape[1])
    ro.G = int(w2.shape[0])
    if ffn is not None:
        st.ffn_weight, st.ffn_bias = _lib.ptr(params[k + 6]), _lib.ptr(params[k + 7])
        st.head_dropout = float(head_dropout)
        st.rng_state, st.rng_used = _lib.ptr(rng), _lib.ptr(rng_used)
    if edge_stats is not None and update_running:
        from .readout import _bn_stats_struct
        es, es_keep = _bn_stats_struct(*edge_stats)
        st.edge_stats = C.pointer(es)
        keep.append((es, es_keep))
    return st, keep


def _batch_struct(plan: BatchPlan, mp: MoleculePlan, x: torch.Tensor):
    b = _lib.MoleculeBatch()
    b.n_atoms, b.n_mols, b.n_chunks, b.max_chunk_atoms = plan.n_atoms, mp.n_mols, mp.n_chunks, mp.max_chunk_atoms
    b.chunk_mol_ptr, b.mol_atom_ptr = mp.chunk_ptr.data_ptr(), mp.mol_ptr.data_ptr()
    b.atom_degree, b.atom_rank = mp.atom_deg.data_ptr(), mp.atom_rank.data_ptr()
    keep = []
    for i, bk in enumerate(plan.buckets):
        d = b.buckets[i]
        d.count = bk.count
        if bk.count:
            eu = bk.e_unit(8 if bk.e_nei is None else int(bk.e_nei.shape[-1]))
            keep.append(eu)
            d.selected_index, d.nei_index, d.nei_edge_unit = bk.sel.data_ptr(), bk.nei.data_ptr(), eu.data_ptr()
            d.p_focal, d.nei_p = _lib.ptr(bk.p_focal), _lib.ptr(bk.nei_p)
    b.x, b.x_stride = x.data_ptr(), int(x.stride(0))
    return b, keep


class _Arena:
    """Slices of ONE device buffer (16-byte aligned), handed out in order: an eager step at these sizes is host-bound, and ~60
    small ``torch.empty`` calls were a tenth of it."""

    def __init__(self, dev):
        self.dev, self.want, self.buf, self.off = dev, [], None, 0

    def plan(self, numel: int) -> int:
        self.want.append(numel)
        return len(self.want) - 1

    def allocate(self):
        total = sum((n + 3) // 4 * 4 for n in self.want)
        self.buf = torch.empty(max(total, 4), dtype=torch.float32, device=self.dev)
        offs, o = [], 0
        for n in self.want:
            offs.append(o)
            o += (n + 3) // 4 * 4
        self.offs = offs

    def take(self, slot: int, shape):
        n = self.want[slot]
        return self.buf[self.offs[slot]:self.offs[slot] + n].view(shape)


def _alloc_state(net, plan: BatchPlan, params, want_grads: bool, want_sims: bool):
    """Pair records (+ the last layer's chirality record) of every layer, and -- for a backward -- the gradient tensors
    (views of one buffer each: the pair records of a call, the kernel-bank gradients of a call)."""
    dev = plan.device
    layers = net.gnn.layers
    rec, gra = _Arena(dev), _Arena(dev)
    todo = []
    k = 0
    for li, layer in enumerate(layers):
        lp = params[k:k + 28]
        k += 28
        for d in range(4):
            L, nd = int(lp[7 * d].shape[0]), plan.buckets[d].count
            if L == 0 or nd == 0:
                todo.append(None)
                continue
            last4 = d == 3 and li == len(layers) - 1
            ent = [li, d, nd, L, rec.plan(nd * L * 4), rec.plan((nd * L + 3) // 4) if last4 else None, None]
            if want_grads:
                ent[6] = (gra.plan(lp[7 * d].numel()), gra.plan(lp[7 * d + 1].numel()), gra.plan(lp[7 * d + 2].numel()), gra.plan(4))
            todo.append((ent, lp))
    rec.allocate()
    if want_grads:
        gra.allocate()
    saved = [[None] * 4 for _ in layers]
    grads = [[None] * 4 for _ in layers]
    for item in todo:
        if item is None:
            continue
        (li, d, nd, L, s_pr, s_ch, s_g), lp = item
        pr = rec.take(s_pr, (nd, L, 4))
        ch = None
        if s_ch is not None:
            ch = rec.take(s_ch, ((nd * L + 3) // 4,)).view(torch.int8)[:nd * L].view(nd, L)
        saved[li][d] = (pr, ch)
        if s_g is not None:
            grads[li][d] = (gra.take(s_g[0], lp[7 * d].shape), gra.take(s_g[1], lp[7 * d + 1].shape),
                            gra.take(s_g[2], lp[7 * d + 2].shape), gra.take(s_g[3], (4,)))
    sims = []
    k = 0
    for li in range(len(layers)):
        lp = params[k:k + 28]
        k += 28
        K = sum(int(lp[7 * d].shape[0]) for d in range(4))
        sims.append(torch.zeros((plan.n_atoms, K), dtype=torch.float32, device=dev) if want_sims else None)
    return saved, grads, sims


def _run(net, ffn, params, plan, mp, x, mode, target, grad_emb, head_dropout, update_running, edge_stats=None):
    lib = _lib.load()
    dev = x.device
    want_grads = bool(mode & _lib.MOLECULE_BACKWARD)
    cap = debug_capture
    saved, grads, sims = _alloc_state(net, plan, params, want_grads, cap is not None)
    small = {}
    k = 28 * len(net.gnn.layers)
    if want_grads:
        for name, p in zip(("bn_w", "bn_b", "w1", "b1", "w2", "b2"), params[k:k + 6]):
            small[name] = None if p is None else torch.empty_like(p)
        if ffn is not None:
            small["ffn_w"] = torch.empty_like(params[k + 6])
            small["ffn_b"] = None if params[k + 7] is None else torch.empty_like(params[k + 7])
    rng = used = None
    if ffn is not None and head_dropout > 0.0:
        from .readout import head_rng_state
        rng = head_rng_state(dev)
        used = torch.empty(2, dtype=torch.int64, device=dev)
    st, keep = _net_struct(net, ffn, params, grads if want_grads else None, saved, sims if cap is not None else None,
                           head_dropout, update_running, rng, used, edge_stats)
    if want_grads:
        st.grad_bn_weight, st.grad_bn_bias = _lib.ptr(small["bn_w"]), _lib.ptr(small["bn_b"])
        st.grad_lin1_weight, st.grad_lin1_bias = _lib.ptr(small["w1"]), _lib.ptr(small["b1"])
        st.grad_lin2_weight, st.grad_lin2_bias = _lib.ptr(small["w2"]), _lib.ptr(small["b2"])
        if ffn is not None:
            st.grad_ffn_weight, st.grad_ffn_bias = _lib.ptr(small["ffn_w"]), _lib.ptr(small.get("ffn_b"))
    bs, keep_b = _batch_struct(plan, mp, x)
    G = int(params[k + 4].shape[0])
    emb = torch.empty((mp.n_mols, G), dtype=torch.float32, device=dev)
    pred = loss = None
    if mode & _lib.MOLECULE_HEAD:
        pred = torch.empty(mp.n_mols, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nbytes = int(lib.mkgnn_molecule_workspace_bytes(C.byref(st), int(x.shape[1]), plan.n_atoms, mp.n_chunks))
        if nbytes == 0:
            raise _lib.MolKGNNLibraryError("mkgnn_molecule_workspace_bytes: model shape not covered")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.mkgnn_molecule_step(C.byref(st), C.byref(bs), int(mode), _lib.ptr(target), _lib.ptr(grad_emb),
                                           emb.data_ptr(), _lib.ptr(pred), _lib.ptr(loss), ws.data_ptr(), ws.numel(),
                                           _lib.stream_ptr(dev)), "mkgnn_molecule_step")
    if cap is not None:
        cap.update(saved=saved, sims=sims, emb=emb, pred=pred)
    # gradients in the order of flat_parameters (None where the reference's autograd leaves none: p_support, banks of a
    # degree absent from the batch)
    flat = None
    if want_grads:
        flat = []
        for li in range(len(net.gnn.layers)):
            for d in range(4):
                g = grads[li][d]
                if g is None:
                    flat += [None] * 7
                else:
                    flat += [g[0], g[1], g[2], None, g[3][0], g[3][1], g[3][2]]
        flat += [small["bn_w"], small["bn_b"], small["w1"], small["b1"], small["w2"], small["b2"]]
        if ffn is not None:
            flat += [small["ffn_w"], small.get("ffn_b")]
    del keep, keep_b
    return emb, pred, loss, flat


def _shape_like(g, p):
    return None if g is None else g.reshape(p.shape)


class _MoleculeNetFn(torch.autograd.Function):
    """``MolKGNNNet.forward`` as one operator: graph embedding out; the backward recomputes the forward inside the same
    launch that runs the backward (nothing but the parameters is kept)."""

    @staticmethod
    def forward(ctx, net, plan, mp, es, x, *params):
        params = list(params)
        emb, _, _, _ = _run(net, None, params, plan, mp, x, 0, None, None, 0.0, True, es)
        ctx.net, ctx.plan, ctx.mp, ctx.x, ctx.params = net, plan, mp, x, params
        return emb

    @staticmethod
    def backward(ctx, grad_emb):
        g = grad_emb.contiguous().float()
        _, _, _, flat = _run(ctx.net, None, ctx.params, ctx.plan, ctx.mp, ctx.x,
                             _lib.MOLECULE_BACKWARD | _lib.MOLECULE_GRAD_EMB, None, g, 0.0, False)
        return (None, None, None, None, None) + tuple(_shape_like(gr, p) if p is not None else None for gr, p in zip(flat, ctx.params))


class _MoleculeLossFn(torch.autograd.Function):
    """``GNNModel.loss`` as one operator: forward, BCE head and -- when a gradient will be asked for -- the whole backward in
    the same launch (d loss = 1; any other incoming gradient scales the stored ones)."""

    @staticmethod
    def forward(ctx, net, ffn, p_drop, target, plan, mp, es, x, *params):
        params = list(params)
        need = any(ctx.needs_input_grad[8:])
        mode = _lib.MOLECULE_HEAD | (_lib.MOLECULE_BACKWARD if need else 0)
        _, pred, loss, flat = _run(net, ffn, params, plan, mp, x, mode, target.reshape(-1).float().contiguous(), None, p_drop, True, es)
        ctx.flat, ctx.params = flat, params
        ctx.pred = pred
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        from .readout import _is_unit_seed
        flat = ctx.flat
        if flat is None:
            raise RuntimeError("molecule-resident loss: no gradient was computed in the forward")
        if not _is_unit_seed(grad_loss):
            gl = grad_loss.reshape(()).float()
            flat = [None if g is None else g * gl for g in flat]
        return (None,) * 8 + tuple(_shape_like(g, p) if p is not None else None for g, p in zip(flat, ctx.params))


def _plan_of(data) -> BatchPlan:
    from .plan import plan_from_lists_cached
    names = ('p_focal', 'nei_p', 'nei_edge_attr', 'selected_index', 'nei_index')
    units = [getattr(data, f'nei_edge_unit_deg{d}', None) for d in range(1, 5)]
    return plan_from_lists_cached(data.x.shape[0], *[[getattr(data, f'{nm}_deg{d}') for d in range(1, 5)] for nm in names],
                                  data.edge_index, units if any(u is not None for u in units) else None)


def _ready(net, data, ffn):
    """(plan, molecule plan, parameters) when the molecule-resident path takes this model and batch, else None."""
    if _MODE == "0" or not data.x.is_cuda:
        return None
    n_mols = getattr(data, 'num_graphs', None)
    if n_mols is not None and not wanted(int(n_mols)):
        return None
    if not model_qualifies(net, data, ffn):
        return None
    if getattr(data, '_rf_ready', None) is not None:     # degree buckets still being filled on the index stream: join first
        from .receptive_field import await_receptive_fields
        await_receptive_fields(data)
    plan = _plan_of(data)
    if _MODE != "1" and n_mols is not None and int(n_mols) > _MAX_MOLS_AUTO and not torch.cuda.is_current_stream_capturing() \
            and not getattr(data, "resident", False):
        # An eager step above the captured threshold.  On RESIDENT batches the one-launch step is 1.7x the faster eager step
        # (1.07 against 1.84 ms at 64-256 molecules, round 5); on a batch seen for the FIRST time its chunk table has to be built
        # (sorts over the edge list + one host round trip) and the step is no faster, at 512 molecules slower (4.3 against
        # 3.1 ms: tools/diag/eager_fresh_probe.py) -- and a training loop that streams its data sees every batch once.
        # Round 5 let the visit count decide (first sight per operator, the one-launch step when a batch came back): the same
        # batch then ran through two kernel families in epochs 1 and 2, whose last-bit differences flip tied neighbour orders
        #  -- a run was not reproducible from its inputs.  Now the choice is a function of the
        # batch alone: the caller that KEEPS its batches says so (``data.resident = True``: the one-launch step from the first
        # visit on, chunk table built once), everybody else gets the per-operator kernels on every visit.
        return None
    mp = molecule_plan(plan, getattr(data, 'batch', None), n_mols)
    if mp is None or not wanted(mp.n_mols):
        return None
    if mp.n_mols > _MAX_MOLS_AUTO and _MODE != "1" and not torch.cuda.is_current_stream_capturing():
        # This batch runs here only while it is launched eagerly; once the step is CAPTURED it takes the per-operator
        # path (wanted()), and that path's lazy per-batch builds synchronise with the host (readout.MoleculeSegments).
        # Build them now, outside any capture, so the captured step is not the per-operator path's first use of the batch.
        _prebuild_per_operator(data, mp.n_mols)
    return plan, mp, flat_parameters(net, ffn)


def _prebuild_per_operator(data, n_mols: int) -> None:
    if getattr(data, 'mol_ptr', None) is not None and getattr(data, 'atom_mol', None) is not None:
        return                                           # (segments come with the batch: nothing is built lazily)
    bv = getattr(data, 'batch', None)
    if bv is not None:
        from .readout import molecule_segments
        molecule_segments(bv, getattr(data, 'num_graphs', None))


def net_forward(net, data) -> Optional[torch.Tensor]:
    """``MolKGNNNet.forward(data)`` through the molecule-resident kernels, or ``None`` if they do not take it.  Only with
    ``MKGNN_MOLECULE=1``: the forward alone is the slower one even at 16 molecules (0.084 against 0.074 ms per operator, bench.py
    ``small_batch.*.paths.forward_only_ms``); what the one-launch form wins it wins in the backward, i.e. through ``loss_forward``."""
    if _MODE != "1":
        return None
    r = _ready(net, data, None)
    if r is None:
        return None
    plan, mp, params = r
    es = net._edge_stats(data) if hasattr(net, '_edge_stats') else None
    if es is not None and not _edge_stats_in_launch(es):
        from .readout import update_running_stats
        update_running_stats(*es)
        es = None
    return _MoleculeNetFn.apply(net, plan, mp, es, data.x, *params)


def loss_forward(model, data, p_drop: float) -> Optional[torch.Tensor]:
    """``GNNModel.loss(data)`` (single task, BCE with logits) through the molecule-resident kernels, or ``None``."""
    net, ffn = model.gnn_model, model.ffn
    r = _ready(net, data, ffn)
    if r is None or data.y.numel() != r[1].n_mols:
        return None
    plan, mp, params = r
    es = net._edge_stats(data) if hasattr(net, '_edge_stats') else None
    # edge_batch_norm's side effect (reference MolKGNNNet.py:116): one more block of the step's preparation launch, or -- more
    # bond rows than one block takes -- a small launch of its own
    if es is not None and not _edge_stats_in_launch(es):
        from .readout import update_running_stats
        update_running_stats(*es)
        es = None
    return _MoleculeLossFn.apply(net, ffn, float(p_drop), data.y, plan, mp, es, data.x, *params)
