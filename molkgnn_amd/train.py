"""The training step the throughput metric is quoted on: the reference's ``GNNModel``
(``model.py:127-198``) around the HIP kernel convolution, without Lightning.

Only what a forward + backward (+ optimiser) step needs is restated here: the module tree with the
reference's parameter names (``gnn_model.*``, ``lin1``, ``lin2``, ``ffn``; 132 300 parameters), the
``BCEWithLogitsLoss`` the data module selects (``data.py:37``) and the AdamW groups chosen by parameter
name (``model.py:373-382``).  Logging, checkpoints, metrics and the LR schedule are out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.nn import BCEWithLogitsLoss, Dropout, Linear, ReLU

from .MolKGNNNet import MolKGNNNet

KERNEL_COUNTS = (10, 20, 30, 50)    # paper / README kernels per degree


class GNNModel(torch.nn.Module):
    def __init__(self, num_layers=3, kernels_1hop=KERNEL_COUNTS, kernels_Nhop=KERNEL_COUNTS, node_feature_dim=28,
                 edge_feature_dim=7, hidden_dim=32, dropout_ratio=0.0, ffn_dropout_rate=0.25, ffn_hidden_dim=64,
                 task_dim=1):
        super().__init__()
        kw = {f"num_kernel{d}_1hop": k for d, k in zip(range(1, 5), kernels_1hop)}
        kw.update({f"num_kernel{d}_Nhop": k for d, k in zip(range(1, 5), kernels_Nhop)})
        self.gnn_model = MolKGNNNet(num_layers=num_layers, x_dim=node_feature_dim, edge_attr_dim=edge_feature_dim,
                                    graph_embedding_dim=hidden_dim, drop_ratio=dropout_ratio, **kw)
        self.lin1 = Linear(hidden_dim, ffn_hidden_dim)     # unused by forward, as in the reference (model.py:147-148)
        self.lin2 = Linear(ffn_hidden_dim, task_dim)
        self.ffn = Linear(hidden_dim, task_dim)
        self.dropout = Dropout(p=ffn_dropout_rate)
        self.activate_func = ReLU()
        self.loss_func = BCEWithLogitsLoss()

    def forward(self, data):
        graph_embedding = self.dropout(self.gnn_model(data))
        if self.ffn.out_features == 1 and self.ffn.bias is not None and graph_embedding.dim() == 2:
            # ffn(graph_embedding) for a single task as multiply + row sum: the [1 x B] @ [B x H] weight gradient of
            # the GEMM form runs a 25 us rocBLAS kernel at B = 4096 (the gemv form 29 us), this takes a few
            pred = (graph_embedding * self.ffn.weight[0]).sum(dim=1, keepdim=True) + self.ffn.bias
        else:
            pred = self.ffn(graph_embedding)
        return pred, graph_embedding

    def loss(self, data):
        if self.ffn.out_features == 1 and type(self.loss_func) is BCEWithLogitsLoss and self.loss_func.reduction == "mean" \
                and self.loss_func.pos_weight is None and self.loss_func.weight is None and data.x.is_cuda:
            from .readout import bce_head_loss
            # dropout -> ffn -> loss in one kernel each way (same formula, 2 kernels instead of ~25; the dropout mask
            # comes from the kernels' own counter-based generator, see readout.head_rng_state)
            p = self.dropout.p if (self.training and self.dropout.p < 1.0) else 0.0
            if not (self.training and self.dropout.p >= 1.0):
                # small batches: forward, loss and (when a gradient will be asked for) the whole backward in one launch
                from . import molecule as _mol
                fused = _mol.loss_forward(self, data, p)
                if fused is not None:
                    return fused
            nreal = getattr(data, 'n_valid_molecules', None)
            # large batches: everything behind the last convolution -- readout, head, loss and their gradients -- in one launch
            # (readout.tail_loss) where the model and the batch qualify; the embedding otherwise
            tail = None if (self.training and self.dropout.p >= 1.0) else (self.ffn, data.y, p, nreal)
            graph_embedding = self.gnn_model(data, _tail=tail)
            if isinstance(graph_embedding, tuple):
                return graph_embedding[1]
            if self.training and self.dropout.p >= 1.0:
                graph_embedding = self.dropout(graph_embedding)
            # (a padded batch: the padding molecules' rows take no part in the loss -- the head reads the first nreal rows)
            return bce_head_loss(graph_embedding, self.ffn, data.y, dropout_p=p, n_rows=nreal)
        pred, _ = self(data)
        return self.loss_func(pred.view(-1), data.y.view(-1).float())


_ONES: dict = {}


def backward(loss: torch.Tensor) -> None:
    """``loss.backward()`` with the kernel-bank gradients of every KernelSetConv left running beside the layers below
    (functional.deferred_bank_gradients); all of them are complete, in stream order, when this returns -- which is
    all the optimiser (or a gradient all-reduce) that follows needs.  Use it where parameters' ``.grad`` start out as
    None (``zero_grad(set_to_none=True)``); calls that would accumulate are simply not deferred."""
    from .functional import deferred_bank_gradients
    # d loss / d loss = 1 from a tensor that already exists (autograd otherwise fills a fresh one: a launch per step)
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():
            one = None                        # (first use inside a capture: let autograd make its own this once)
        else:
            one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
            from .readout import register_unit_gradient
            register_unit_gradient(one)       # the head's fused forward has d loss = 1 gradients ready: nothing to launch
    with deferred_bank_gradients():
        if one is not None and loss.dim() == 0:
            loss.backward(one)
        else:
            loss.backward()


def training_step(model, batch, optimizer=None) -> torch.Tensor:
    """``loss = model.loss(batch); backward(loss); optimizer.step()`` as ONE unit (gradients start out as None): what a training loop
    does per batch, with the licence that gives -- nothing reads the loss or a parameter gradient between the forward and the
    optimiser, so the fused tail's last reduction leaves the critical chain (``readout.deferred_tail_reduce``).  Returns the loss
    (complete, in stream order, when this returns).  Same kernels, same arithmetic, same bits as the three calls."""
    from .readout import deferred_tail_reduce
    model.zero_grad(set_to_none=True)
    with deferred_tail_reduce(batch.x.device if batch.x.is_cuda else None):
        loss = model.loss(batch)
        backward(loss)
    if optimizer is not None:
        optimizer.step()
    return loss


def tune_torch_backends() -> None:
    """PyTorch-side knobs for the plain-PyTorch parts of the step (readout GEMMs).  The weight-gradient GEMM
    [32 x N] @ [N x 110] (N ~ 1e5) takes ~225 us through hipBLASLt and ~53 us through rocBLAS on MI355X."""
    try:
        torch.backends.cuda.preferred_blas_library("cublas")     # = rocBLAS on ROCm
    except Exception:
        pass


def configure_optimizer(model: torch.nn.Module, weight_decay: float = 0.0, lr: float = 1e-3, fused: Optional[bool] = None,
                        capturable: bool = False):
    """AdamW with the kernel parameters exempt from weight decay, selected by name (model.py:373-382)."""
    decay, nodecay = [], []
    for name, p in model.named_parameters():
        if ('x_center' in name) or ('p_support' in name) or (
                ('edge_attr_support' in name) and ('edge_attr_support_sc' not in name)) or ('x_support' in name):
            nodecay.append(p)
        else:
            decay.append(p)
    groups = [{'params': nodecay, 'weight_decay': 0}, {'params': decay, 'weight_decay': weight_decay}]
    if fused is None:
        fused = all(p.is_cuda and p.dtype == torch.float32 for p in model.parameters())
    if fused:
        from .optim import FusedAdamW     # one HIP launch for the whole model; step counters on the device (capturable)
        opt = FusedAdamW(groups, lr=lr)
        opt.prepare_state()               # state exists before anything can be captured (a fill inside a graph would replay)
        return opt
    kw = {}
    if capturable:
        kw["capturable"] = True          # step counters live on the device: the step can be replayed from a hipGraph
    return torch.optim.AdamW(groups, lr=lr, **kw)


class CapturedSteps:
    """``loss = steps(batch)``: one training step (``model.loss`` -> ``backward`` -> ``optimizer.step``) per call, launched eagerly
    the first ``warmup`` times a batch is seen and from a hipGraph captured on its next visit afterwards.

    The reference's harness launches every operator of every step from Python (Lightning's loop around ``model.py``); with the
    HIP operators that costs 2-3 ms of host time per step at batch 16-256, against 0.25-0.28 ms of GPU work (bench.py,
    ``small_batch.*.paths.eager_ms_per_step``).  A loop that revisits its batches -- several epochs over resident data -- gets the
    replayed step with two lines::

        steps = CapturedSteps(model, optimizer)
        for epoch in range(E):
            for batch in resident_batches:
                loss = steps(batch)            # a 0-d tensor, valid until the next call with the SAME batch

    A batch is recognised by the identity of its object and of its ``x`` storage; it must stay alive and unchanged (a batch whose
    tensors are refilled in place belongs in ``padding.StaticBatch``, which is built for that).  Each captured batch keeps its own
    graph (and the memory of one step's activations): ``max_graphs`` bounds their number, batches beyond it stay eager.
    ``optimizer`` must be capturable (``configure_optimizer(..., capturable=True)`` or the fused AdamW)."""

    def __init__(self, model: torch.nn.Module, optimizer=None, warmup: int = 2, max_graphs: int = 64):
        self.model, self.optimizer = model, optimizer
        self.warmup, self.max_graphs = int(warmup), int(max_graphs)
        self._seen: dict = {}
        self._graphs: dict = {}
        self._stream = None

    def _key(self, batch):
        return (id(batch), batch.x.data_ptr(), batch.x._version)

    def _eager(self, batch):
        return training_step(self.model, batch, self.optimizer).detach()

    def __call__(self, batch):
        key = self._key(batch)
        hit = self._graphs.get(key)
        if hit is not None:
            graph, static_loss, _keep = hit
            graph.replay()
            return static_loss
        n = self._seen.get(key, 0)
        self._seen[key] = n + 1
        if n < self.warmup or len(self._graphs) >= self.max_graphs or not batch.x.is_cuda:
            return self._eager(batch)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=batch.x.device)
        side = self._stream
        side.wait_stream(torch.cuda.current_stream(batch.x.device))
        with torch.cuda.stream(side):
            self.model.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                static_loss = training_step(self.model, batch, self.optimizer).detach()
        torch.cuda.current_stream(batch.x.device).wait_stream(side)
        self._graphs[key] = (graph, static_loss, batch)      # (the batch object is kept: its id stays its own)
        graph.replay()                                        # (a capture launches nothing: this visit's step)
        return static_loss
