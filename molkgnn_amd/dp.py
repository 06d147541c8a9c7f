"""Data parallelism for the kernel-convolution model: one process per GPU, molecules sharded
across ranks, one gradient all-reduce per step over RCCL/xGMI (backend "nccl" on ROCm).

The reference has no distributed code; a batch is a disjoint union of
molecules and nothing on the path crosses a molecule boundary, so the only exchange is the
gradient sum.  The model is tiny (132 300 floats, 0.53 MB): the collective is latency bound,
so every gradient travels in ONE flat fp32 buffer and one ``all_reduce`` per step -- bucketing
or overlap would only add launches.

Some parameters never receive a gradient (``p_support``, the two unused score weights, the
unused heads) and a degree that is absent from a rank's batch leaves its
bank's gradients ``None`` on that rank only.  The flat buffer therefore has a fixed slot for
every parameter that CAN receive a gradient, zero-filled where this rank has none, so all
ranks always reduce the same layout.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None) -> int:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return world


def shard_indices(n_items: int, rank: int, world: int) -> range:
    """Batch ids of this rank: ``rank, rank + world, ...`` -- the union over ranks is the 1-GPU stream."""
    return range(rank, n_items, world)


class FlatGradAllReduce:
    """Sum-then-average the gradients of ``params`` across ranks through one flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], never_trained: Iterable[str] = (), names=None):
        params = list(params)
        names = list(names) if names is not None else [str(i) for i in range(len(params))]
        skip = tuple(never_trained)

        def skipped(name: str) -> bool:
            # "^prefix" matches the start of the parameter name, anything else matches anywhere in it
            return any(name.startswith(s[1:]) if s.startswith("^") else s in name for s in skip)
        self.params: List[torch.nn.Parameter] = [p for p, n in zip(params, names) if p.requires_grad and not skipped(n)]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._filled: List[torch.nn.Parameter] = []

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    def grads(self) -> List[Optional[torch.Tensor]]:
        """The current ``p.grad`` tensors of the reduced parameters, in buffer order.  A step replayed from a captured
        graph writes its gradients into the tensors that were ``p.grad`` *when that graph was captured*; with one graph
        per resident batch those are different tensors per graph, so the caller keeps this list per graph and hands
        it to ``reduce`` (``p.grad`` itself only names the last captured graph's tensors)."""
        return [p.grad for p in self.params]

    def sum_into_flat(self, grads: List[Optional[torch.Tensor]]) -> None:
        """Copy the given gradients (from ``grads()``) into the flat buffer and SUM it over the ranks; nothing is
        scaled or copied back.  For an optimiser that reads the flat views directly and divides by the world size
        itself (``FusedAdamW(grad_scale=1 / world)`` captured with ``p.grad = view``): two copy launches and one
        scaling launch fewer per step than ``reduce``."""
        if len(grads) != len(self.params):
            raise ValueError("grads must come from FlatGradAllReduce.grads()")
        have = [(v, g) for v, g in zip(self.views, grads) if g is not None]
        if len(have) != len(self.params):
            self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def reduce(self, grads: Optional[List[Optional[torch.Tensor]]] = None) -> None:
        """Call after backward: afterwards every gradient tensor (``p.grad``, or the given list from ``grads()``) holds
        the mean over ranks."""
        if self.world == 1:
            return
        if grads is not None:
            if len(grads) != len(self.params):
                raise ValueError("grads must come from FlatGradAllReduce.grads()")
            have = [(v, g) for v, g in zip(self.views, grads) if g is not None]
            if len(have) != len(self.params):
                self.flat.zero_()
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)
            if have:
                torch._foreach_copy_([g for _, g in have], [v for v, _ in have])
            return
        for p in self._filled:                       # gradients this object created last time are not this step's
            p.grad = None
        self._filled = []
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        if len(have) != len(self.params):            # slots of parameters without a gradient on this rank contribute zero
            self.flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / self.world)
        if have:                                     # back into the (possibly graph-static) gradient tensors
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])
        for v, p in zip(self.views, self.params):
            if p.grad is None:                       # e.g. a degree absent from this rank's batch
                p.grad = v.clone()
                self._filled.append(p)


# parameter-name fragments that never receive a gradient in the reference's model; "^" anchors a
# prefix: GNNModel's own unused heads are lin1 / lin2 (model.py:147-148), while gnn_model.graph_embedding_lin1 / lin2
# ARE trained and must be reduced
NEVER_TRAINED = ("p_support", "length_sc_weight", "angle_sc_weight", "edge_batch_norm", "graph_embedding_linear",
                 "^lin1.", "^lin2.")
