"""Drop-in for the reference's ``models/MolKGNN/KernelLayer.py``: ``MolGCN``,
the stack of kernel-convolution layers with the neighbour sum between them.

``MolGCN`` keeps the reference's constructor, its keyword-only ``forward`` and
its ``message(sim_sc_j)``; ``propagate`` (PyG ``aggr='add'``,
KernelLayer.py:14,119-123) runs as a CSR segment-sum HIP kernel.
"""
from __future__ import annotations

import os

import torch
from torch.nn import ModuleList

from . import functional as Fn
from .kernels import KernelSetConv
from .plan import plan_from_lists, plan_from_lists_cached
from .receptive_field import GraphBatch

# diagnostics: MKGNN_DENSE_PROPAGATE=1 keeps sim_sc dense (zero-filled rows, dense sums) between convolution and propagate
_BLOCK_ROWS = os.environ.get("MKGNN_DENSE_PROPAGATE") is None
# (measured, round 6: the fork / join around the side stream costs the captured step what the overlap buys -- bank_prepare 5 us
# beside the batch norm, 10 us of gap in front of the first convolution: 0.7307 against 0.7295 ms.  Kept as an opt-in.)
_PREPARE_EARLY = os.environ.get("MKGNN_PREPARE_EARLY", "0") == "1"
# round 6: the preparation is left PENDING in front of a caller's batch norm, whose statistics launch carries its tasks in blocks
# of its own (functional.prepare_banks(defer=True)): the launch leaves the chain in front of the first convolution
_PREPARE_DEFER = os.environ.get("MKGNN_PREPARE_DEFER", "1") != "0"
_FUSE_PROPAGATE = _BLOCK_ROWS and os.environ.get("MKGNN_SPLIT_PROPAGATE") is None
# h between two layers written as the pre-split rows the next layer's matrix instructions take (functional.ROWS_SPLIT, round 6);
# MKGNN_ROWS_SPLIT=0: ordinary fp32 rows (A/B, diagnostics)
_ROWS_SPLIT = _FUSE_PROPAGATE and os.environ.get("MKGNN_ROWS_SPLIT", "1") != "0"
_PREPARE_ONCE = os.environ.get("MKGNN_PREPARE_PER_LAYER") is None

try:
    from torch_geometric.nn import MessagePassing  # type: ignore
except Exception:  # PyG absent: the one aggregation the reference uses
    class MessagePassing(torch.nn.Module):
        """Minimal ``MessagePassing`` (add-aggregation, source -> target)."""

        def __init__(self, aggr='add', **kwargs):
            super().__init__()
            if aggr != 'add':
                raise NotImplementedError("only aggr='add' is provided")
            self.aggr = aggr

        def message(self, **kwargs):
            raise NotImplementedError

try:
    from torch_geometric.data import Data  # type: ignore
except Exception:
    Data = GraphBatch


class MolGCN(MessagePassing):
    def __init__(self, num_layers=5, num_kernel1_1hop=0, num_kernel2_1hop=0, num_kernel3_1hop=0,
                 num_kernel4_1hop=0, num_kernel1_Nhop=0, num_kernel2_Nhop=0, num_kernel3_Nhop=0,
                 num_kernel4_Nhop=0, x_dim=5, p_dim=3, edge_attr_dim=1, ):
        super(MolGCN, self).__init__(aggr='add')
        self.num_layers = num_layers
        if num_layers < 1:
            raise Exception('at least one convolution layer is needed')
        self.layers = ModuleList()
        self.num_kernels_list = []
        # first layer (KernelLayer.py:21-37)
        if (num_kernel1_1hop is not None) and (num_kernel2_1hop is not None) and (
                num_kernel3_1hop is not None) and (num_kernel4_1hop is not None):
            kernel_layer = KernelSetConv(num_kernel1_1hop, num_kernel2_1hop, num_kernel3_1hop, num_kernel4_1hop,
                                         D=p_dim, node_attr_dim=x_dim, edge_attr_dim=edge_attr_dim)
            num_kernels = num_kernel1_1hop + num_kernel2_1hop + num_kernel3_1hop + num_kernel4_1hop
        else:
            raise Exception('MolGCN: num_kernel1-4 need to be specified')
        self.layers.append(kernel_layer)
        self.num_kernels_list.append(num_kernels)
        # N-hop layers (KernelLayer.py:39-48): input width = previous layer's kernel count
        for i in range(num_layers - 1):
            kernel_layer = KernelSetConv(L1=num_kernel1_Nhop, L2=num_kernel2_Nhop, L3=num_kernel3_Nhop,
                                         L4=num_kernel4_Nhop, D=p_dim, node_attr_dim=self.num_kernels(i),
                                         edge_attr_dim=edge_attr_dim)
            self.layers.append(kernel_layer)
            self.num_kernels_list.append(kernel_layer.get_num_kernel())
        self._plan = None

    def num_kernels(self, layer):
        return self.num_kernels_list[layer]

    def prepare_banks_early(self, x, n_slots: int) -> None:
        """Round 6: normalise the kernel banks of all layers NOW, on the device's side stream (``plan.index_stream``), for the
        ``forward`` that follows on rows shaped like ``x`` -- a caller that has something else to run first (MolKGNNNet: the
        batch norm) overlaps the two; ``forward`` joins the stream in front of the first convolution.  A no-op where the
        one-launch preparation does not apply.  Opt-in, ``MKGNN_PREPARE_EARLY=1`` (in a captured step the fork / join costs what
        the overlap buys: see ``_PREPARE_EARLY``)."""
        if x.is_cuda:
            self.drop_pending_prepare(x.device)           # (one an earlier, interrupted forward left behind)
        self._early = None
        if x.is_cuda and _PREPARE_ONCE and _PREPARE_DEFER and not _PREPARE_EARLY and self.num_layers <= 4 \
                and all(layer._can_prepare() for layer in self.layers):
            pl = [layer._bank_params("train", x) for layer in self.layers]
            prepared = Fn.prepare_banks([p for p, _ in pl], [x.shape[1]] + [self.num_kernels(i) for i in range(self.num_layers - 1)],
                                        pl[0][1], x.shape[0], n_slots, defer=True)
            self._early = ((x.shape[0], x.shape[1], x.device), prepared, None)
            return
        if not (x.is_cuda and _PREPARE_ONCE and _PREPARE_EARLY and all(layer._can_prepare() for layer in self.layers)):
            return
        from .plan import index_stream
        side = index_stream(x.device)
        if side is None:
            return
        pl = [layer._bank_params("train", x) for layer in self.layers]
        prepared = Fn.prepare_banks([p for p, _ in pl], [x.shape[1]] + [self.num_kernels(i) for i in range(self.num_layers - 1)],
                                    pl[0][1], x.shape[0], n_slots, side=side)
        self._early = ((x.shape[0], x.shape[1], x.device), prepared, side)

    def drop_pending_prepare(self, device) -> None:
        """An error between ``prepare_banks_early`` and ``forward``: a preparation left pending is withdrawn (nothing else would)."""
        early, self._early = getattr(self, "_early", None), None
        if early is not None and early[2] is None and torch.device(device).type == "cuda":
            Fn.prepare_withdraw(device)

    def set_variant(self, variant: str, backward_variant=None):
        for layer in self.layers:
            layer.variant = variant
            layer.backward_variant = backward_variant

    def propagate(self, edge_index, sim_sc=None, **kwargs):
        """``h[i] = sum_{j -> i} message(sim_sc[j])``; ``message`` is the identity (KernelLayer.py:122-123)."""
        plan = self._plan
        if plan is None or plan.edge_index is not edge_index:
            empty = torch.zeros(0, dtype=torch.long, device=sim_sc.device)
            emptyf = torch.zeros(0, device=sim_sc.device)
            plan = plan_from_lists(sim_sc.shape[0], [emptyf] * 4, [emptyf] * 4, [emptyf] * 4, [empty] * 4, [empty] * 4,
                                   edge_index)
        k = sim_sc.shape[1]
        return Fn.propagate_add(sim_sc, plan, out_pad=(-k) % 4)

    def forward(self, *argv, **kwargv):
        if len(argv) != 0:
            raise Exception('Kernel does not take positional argument, use keyword argument instead. '
                            'e.g. model(data=data)')
        x = kwargv['x']
        edge_index = kwargv['edge_index']
        names = ['p_focal', 'nei_p', 'nei_edge_attr', 'selected_index', 'nei_index']
        fields = {f'{nm}_deg{d}': kwargv[f'{nm}_deg{d}'] for nm in names for d in range(1, 5)}
        data = Data(x=x, p=kwargv['p'], edge_index=edge_index, edge_attr=kwargv['edge_attr'], **fields)
        save_score = kwargv['save_score']
        # one index plan per batch, shared by every layer and by propagate
        # (not part of the reference's signature: unit bond rows that came with the receptive fields, if any)
        units = [kwargv.get(f'nei_edge_unit_deg{d}') for d in range(1, 5)]
        self._plan = plan_from_lists_cached(
            x.shape[0], *[[fields[f'{nm}_deg{d}'] for d in range(1, 5)] for nm in names], edge_index,
            units if any(u is not None for u in units) else None)
        # the kernel banks of ALL layers are normalised by one launch (they depend on the parameters only; layer i reads rows
        # of width F_i = K_{i-1}): MKGNN_PREPARE_PER_LAYER=1 keeps one launch per layer (diagnostics)
        prepared = [None] * self.num_layers
        early, self._early = getattr(self, "_early", None), None
        use_early = early is not None and early[0] == (x.shape[0], x.shape[1], x.device)
        if early is not None and early[2] is None:
            # left pending in front of the caller's batch norm (prepare_banks_early): whatever its statistics launch did not
            # carry is launched now, in front of the first convolution (rows of another shape: dropped, prepared afresh below)
            if use_early:
                Fn.prepare_flush(x.device)
            else:
                Fn.prepare_withdraw(x.device)
        elif use_early:                                  # prepared on the side stream: join it here
            torch.cuda.current_stream(x.device).wait_stream(early[2])
        if use_early:
            prepared = early[1]
        elif x.is_cuda and _PREPARE_ONCE and all(layer._can_prepare() for layer in self.layers):
            n_slots = sum(int(fields[f'nei_index_deg{d}'].numel()) for d in range(1, 5))
            pl = [layer._bank_params("train", x) for layer in self.layers]
            # (a caller that ran something with spare blocks in front -- MolKGNNNet's batch norm -- has had the batch's index
            # arrays read there, functional.touch_hint, and says so on the plan; otherwise this launch reads them)
            done, self._plan._touch_done = getattr(self._plan, "_touch_done", False), False
            with Fn.touch_hint(None if done else Fn.plan_touch_list(self._plan)):
                prepared = Fn.prepare_banks([p for p, _ in pl], [x.shape[1]] + [self.num_kernels(i) for i in range(self.num_layers - 1)],
                                            pl[0][1], x.shape[0], n_slots)
        # (private: MolKGNNNet asks for the LAST layer's block rows instead of h -- its readout projects them before the
        # propagate step, readout.readout_blocks; None is returned where that does not apply and h as usual)
        defer = kwargv.get('_defer_last_propagate') is not None
        h = x
        try:
            for i in range(self.num_layers):
                data.x = h
                is_last_layer = (i == self.num_layers - 1)
                if defer and is_last_layer and _BLOCK_ROWS and not save_score:
                    layer = self.layers[i]
                    sim_sc = layer._run_block_rows(h, self._plan, True, prepared=prepared[i])
                    if sim_sc is not None:
                        kwargv['_defer_last_propagate'].append((sim_sc, self._plan, tuple(layer.L)))
                        return None
                # sim_sc goes nowhere but into propagate: block rows (no zero fill, block-sparse sums both ways)
                # ... and convolution + propagate as one operator where that applies (its backward folds the propagate
                # step's gradient into the kernels' pre-pass); MKGNN_SPLIT_PROPAGATE=1: two operators (diagnostics)
                # (an h that only the next layer reads is written pre-split where that layer takes it: never the last one's,
                # which is handed out)
                split_next = (_ROWS_SPLIT and not is_last_layer and not save_score and h.is_cuda
                              and self.layers[i + 1]._accepts_split_rows(self._plan, h))
                sim_sc, propagated = self.layers[i]._run(h, self._plan, is_last_layer, save_score, block_rows=_BLOCK_ROWS,
                                                         fuse_propagate=_FUSE_PROPAGATE, prepared=prepared[i],
                                                         split_next=split_next)
                h = sim_sc if propagated else self.propagate(edge_index=edge_index, sim_sc=sim_sc)
        finally:
            self._plan = None
        return h

    def message(self, sim_sc_j):
        return sim_sc_j
