"""Consumer of the hot path, kept so that the reference's ``model.py`` can build
it unchanged: BatchNorm(x) -> MolGCN -> lin2(dropout(swish(lin1(h)))) -> add-pool
(reference ``models/MolKGNN/MolKGNNNet.py:10-149``).  ``MolGCN`` is the hot
path; the batch norm in front of it and the readout behind it run as HIP
operators too (``readout.py``).
"""
from __future__ import annotations

import torch
from torch.nn import BatchNorm1d, Dropout, Linear

from .KernelLayer import MolGCN
from . import readout as R


def swish(x):
    return x * torch.sigmoid(x)


def global_add_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    return torch.zeros(size, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, batch, x)


import os as _os

# MKGNN_PROJECT_FIRST: '1' always / '0' never take the block-row readout; default: from _PROJECT_FIRST_ATOMS atoms on (the
# three small passes cost one more dependent launch each way than the two big ones: a loss below a few thousand atoms)
_PROJECT_FIRST = _os.environ.get('MKGNN_PROJECT_FIRST', '')
_PROJECT_FIRST_ATOMS = 16384


class MolKGNNNet(torch.nn.Module):
    def __init__(self, num_layers=1, num_kernel1_1hop=0, num_kernel2_1hop=0, num_kernel3_1hop=0,
                 num_kernel4_1hop=0, num_kernel1_Nhop=0, num_kernel2_Nhop=0, num_kernel3_Nhop=0,
                 num_kernel4_Nhop=0, predefined_kernelsets=True, x_dim=5, p_dim=3, edge_attr_dim=1,
                 drop_ratio=0.25, graph_embedding_dim=5):
        super(MolKGNNNet, self).__init__()
        self.num_layers = num_layers
        self.D = p_dim
        n_hop = num_kernel1_Nhop + num_kernel2_Nhop + num_kernel3_Nhop + num_kernel4_Nhop
        # module creation order = the reference's (MolKGNNNet.py:20-57), so a seeded init draws the same numbers
        self.graph_embedding_linear = Linear(n_hop, graph_embedding_dim)
        self.node_batch_norm = BatchNorm1d(x_dim)
        self.edge_batch_norm = BatchNorm1d(edge_attr_dim)
        self.graph_embedding_lin1 = Linear(n_hop, graph_embedding_dim)
        self.graph_embedding_lin2 = Linear(graph_embedding_dim, graph_embedding_dim)
        self.dropout = Dropout(drop_ratio)
        self.act = swish
        if self.num_layers < 1:
            raise ValueError("GNN_graphpred: Number of GNN layers must be greater than 0.")
        self.gnn = MolGCN(num_layers=num_layers, num_kernel1_1hop=num_kernel1_1hop,
                          num_kernel2_1hop=num_kernel2_1hop, num_kernel3_1hop=num_kernel3_1hop,
                          num_kernel4_1hop=num_kernel4_1hop, num_kernel1_Nhop=num_kernel1_Nhop,
                          num_kernel2_Nhop=num_kernel2_Nhop, num_kernel3_Nhop=num_kernel3_Nhop,
                          num_kernel4_Nhop=num_kernel4_Nhop, x_dim=x_dim, p_dim=p_dim,
                          edge_attr_dim=edge_attr_dim)
        self.pool = global_add_pool

    def save_kernellayer(self, path, time_stamp):
        layers = self.gnn.layers
        print(f'{self.D}D, there are {len(layers)} layers')
        for i, layer in enumerate(layers):
            print(f'saving {i}th layer')
            torch.save(layer.state_dict(), f'{path}/{time_stamp}_{i}th_layer.pth')

    def _edge_stats(self, data):
        """(edge_attr, edge_batch_norm, key, key_limit) for ``readout.update_running_stats`` or None (eval mode, no bonds).  A
        batch padded to a fixed shape counts the bonds of its real atoms only: padding bonds start at padding atoms."""
        bn = self.edge_batch_norm
        ea = getattr(data, 'edge_attr', None)
        if not (bn.training and bn.track_running_stats and bn.running_mean is not None) or ea is None or ea.shape[0] == 0:
            return None
        nv = getattr(data, 'n_valid_atoms', None)
        if nv is None:
            return (ea, bn, None, None)
        src = data.edge_index[0]
        return (ea, bn, src if src.is_contiguous() else src.contiguous(), nv)

    def forward(self, *argv, save_score=False, _tail=None):
        if len(argv) != 1:
            # the reference's 33-positional-argument form reads ``data`` afterwards and cannot work
            # (MolKGNNNet.py:70-89 then :115); only the single-``data`` form is meaningful
            raise ValueError("unmatched number of arguments.")
        data = argv[0]
        # small batches (the reference's own regime, README.md:81): batch norm, every layer and the readout in ONE launch, a
        # workgroup per chunk of whole molecules (molkgnn_amd.molecule); None where the model or the batch does not qualify
        if data.x.is_cuda and not save_score:
            from . import molecule as _mol
            emb = _mol.net_forward(self, data)
            if emb is not None:                          # (edge_batch_norm's statistics moved inside: molecule.net_forward)
                return emb
        # (a batch padded to a fixed shape -- molkgnn_amd.padding -- carries its real atom count and its molecule segments)
        # edge_batch_norm(data.edge_attr) (reference MolKGNNNet.py:116): its output never reaches the kernel convolution
        # (SURVEY 8 a-1), but in training mode the call moves the module's running statistics and num_batches_tracked, which
        # are state-dict contents -- that side effect rides along in the node batch norm's launches (readout.batch_norm)
        kw = {f'{nm}_deg{d}': getattr(data, f'{nm}_deg{d}')
              for nm in ('p_focal', 'nei_p', 'nei_edge_attr', 'selected_index', 'nei_index') for d in range(1, 5)}
        for d in range(1, 5):                          # unit bond rows built with the receptive fields (mkgnn_rf_fill), if any
            u = getattr(data, f'nei_edge_unit_deg{d}', None)
            if u is not None:
                kw[f'nei_edge_unit_deg{d}'] = u
        # the normalised features go nowhere but into the first kernel convolution: where that layer takes pre-split rows
        # (functional.ROWS_SPLIT) the batch norm writes them so -- asked from shapes alone, nothing is launched or awaited
        split_x, plan0 = False, None
        if data.x.is_cuda and not save_score:
            from . import KernelLayer as _KL
            from .plan import plan_from_lists_cached
            names = ('p_focal', 'nei_p', 'nei_edge_attr', 'selected_index', 'nei_index')
            units = [kw.get(f'nei_edge_unit_deg{d}') for d in range(1, 5)]
            plan0 = plan_from_lists_cached(data.x.shape[0], *[[kw[f'{nm}_deg{d}'] for d in range(1, 5)] for nm in names],
                                           data.edge_index, units if any(u is not None for u in units) else None)
            # (the banks depend on the parameters only: their one launch can run beside the batch norm, on the side stream)
            self.gnn.prepare_banks_early(data.x, sum(int(kw[f'nei_index_deg{d}'].numel()) for d in range(1, 5)))
            if _KL._ROWS_SPLIT:
                split_x = self.gnn.layers[0]._accepts_split_rows(plan0, data.x)
        try:                                             # (a bank preparation may be pending from here to self.gnn: see except)
            # (round 6: spare blocks of the batch norm's statistics launch read the batch's index arrays once -- cold after the
            # previous step's backward, and otherwise paid for by the first convolution: functional.touch_hint)
            from . import functional as _Fn
            with _Fn.touch_hint(_Fn.plan_touch_list(plan0) if plan0 is not None else None) as hint:
                x = R.batch_norm(data.x, self.node_batch_norm, getattr(data, 'n_valid_atoms', None), companion=self._edge_stats(data),
                                 split_out=split_x)
            if plan0 is not None:
                plan0._touch_done = hint.taken
            if getattr(data, '_rf_ready', None) is not None:     # degree buckets still being built on the index stream
                from .receptive_field import await_receptive_fields
                await_receptive_fields(data)
            seg = None
            if getattr(data, 'mol_ptr', None) is not None and getattr(data, 'atom_mol', None) is not None:
                seg = R.MoleculeSegments.from_tensors(data.mol_ptr, data.atom_mol, getattr(data, 'max_mol_atoms', None),
                                                      getattr(data, 'max_mol_edges', None))
            # The last layer's output goes nowhere but through propagate into lin1: where it applies (large batches: it trades
            # two big passes for three small ones), the readout takes the last convolution's BLOCK ROWS and projects them
            # before the propagate step (readout.readout_blocks); MKGNN_PROJECT_FIRST=0 / 1 forces the choice (diagnostics)
            blocks_out = []
            lin1, lin2 = self.graph_embedding_lin1, self.graph_embedding_lin2
            Ls = self.gnn.layers[-1].L
            dims = (sum(Ls), lin1.weight.shape[0], lin2.weight.shape[0])
            # (from _PROJECT_FIRST_ATOMS atoms on -- or at any size where the dense readout kernels do not take the shape, e.g. 160
            # kernels per layer: the block-row form takes up to 255 columns and keeps such a model off the PyTorch-operator path)
            # (... or at any size when the caller wants the loss itself: the fused tail -- readout.tail_loss -- starts from block rows)
            no_drop = self.dropout is None or not self.dropout.training or self.dropout.p == 0.0
            want_tail = (_tail is not None and R._FUSED_TAIL and torch.is_grad_enabled() and no_drop and x.is_cuda
                         and R.tail_supported(*dims, Ls))
            want = x.is_cuda and not save_score and (_PROJECT_FIRST == '1' or (_PROJECT_FIRST != '0' and (
                x.shape[0] >= _PROJECT_FIRST_ATOMS or not R.readout_supported(*dims) or want_tail)))
            if want:
                want = R.readout_blocks_supported(*dims, Ls)
            if want and seg is None:
                seg = R.molecule_segments(data.batch, getattr(data, 'num_graphs', None))
            want = want and seg.sorted and seg.size > 0
            node_representation = self.gnn(x=x, edge_index=data.edge_index, edge_attr=data.edge_attr, p=data.p,
                                           save_score=save_score, **kw, **({'_defer_last_propagate': blocks_out} if want else {}))
        except BaseException:
            # the deferred bank preparation (MolGCN.prepare_banks_early) must not outlive this call: its workspaces would
            self.gnn.drop_pending_prepare(data.x.device)
            raise
        if node_representation is None:                     # the last propagate was left to the readout
            sim_sc, plan, Ls = blocks_out[0]
            # (private: train.GNNModel.loss asks for the loss itself -- readout, head, loss and all their gradients in one
            # launch, readout.tail_loss -- where that applies; it gets ("loss", value) back, or the embedding as usual)
            if want_tail and sim_sc.requires_grad and R._tail_limits_ok(seg, plan):
                ffn, target, p_head, n_rows = _tail
                return ("loss", R.tail_loss(sim_sc, plan, Ls, lin1, lin2, ffn, target, seg, p_head, n_rows))
            return R.readout_blocks(sim_sc, plan, Ls, lin1, lin2, self.dropout, seg)
        # pool(lin2(dropout(act(lin1(h)))), batch) -- MolKGNNNet.py:144-146 -- as one operator
        return R.readout(node_representation, lin1, lin2, self.dropout, data.batch, getattr(data, 'num_graphs', None), segments=seg)

    @staticmethod
    def add_model_specific_args(parent_parser):
        parser = parent_parser.add_argument_group("MolKGNNNet")
        parser.add_argument('--num_layers', type=int, default=4)
        for hop in ('1hop', 'Nhop'):
            for d, dflt in zip(range(1, 5), (10, 20, 30, 50)):
                parser.add_argument(f'--num_kernel{d}_{hop}', type=int, default=dflt)
        parser.add_argument('--node_feature_dim', type=int, default=28)
        parser.add_argument('--edge_feature_dim', type=int, default=7)
        parser.add_argument('--hidden_dim', type=int, default=32)
        parser.add_argument('--dropout_ratio', type=float, default=0)
        return parent_parser
