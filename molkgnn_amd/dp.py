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
ranks always reduce the same layout; a flag per parameter rides along so that every rank knows
which parameters had a gradient somewhere.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None, force: bool = False,
                                device: Optional[torch.device] = None) -> int:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns world size.
    ``force``: also for a world of one (a rehearsal of the collective path on a single GPU).  ``device``: this rank's
    GPU, handed to the RCCL backend as ``device_id`` (the communicator is bound to it at once instead of guessing the
    device at the first collective)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if world > 1 and "RANK" not in os.environ:
            # a mis-launched multi-rank job: every process would call itself rank 0 and hang in the rendezvous
            raise KeyError("RANK is not set although WORLD_SIZE > 1 (launch with torch.distributed.run)")
        extra = {"device_id": device} if (device is not None and device.type == "cuda" and backend == "nccl") else {}
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")), world_size=world, **extra)
    return world


def shard_indices(n_items: int, rank: int, world: int) -> range:
    """Batch ids of this rank: ``rank, rank + world, ...`` -- the union over ranks is the 1-GPU stream."""
    return range(rank, n_items, world)


class FlatGradAllReduce:
    """Sum-then-average the gradients of ``params`` across ranks through one flat buffer.

    The buffer is ``[gradient slots | one "has a gradient" flag per parameter]``; both halves travel in the same
    all-reduce, so afterwards ``flags[i]`` is the number of ranks that had a gradient for parameter ``i``.  A parameter
    without a gradient on this rank (a degree absent from this rank's batch) still receives the reduced gradient of the
    ranks that had one -- otherwise the replicas drift apart -- and a parameter without a gradient on EVERY rank stays
    without one, as on one GPU (the reference's AdamW skips it: no step count, no moment decay)."""

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
        self._buf = torch.zeros(total + len(self.params), dtype=torch.float32, device=dev)
        self.flat = self._buf[:total]                 # the gradient slots
        self.flags = self._buf[total:]                # ranks that had a gradient, per parameter (after a reduction)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._filled: List[torch.nn.Parameter] = []
        self._flag_cache: dict = {}
        self._zero_cache: dict = {}
        self._copy_tables: dict = {}             # mkgnn_flat_copy tables, by the pointers they hold

    @property
    def nbytes(self) -> int:
        return self._buf.numel() * 4

    def grads(self) -> List[Optional[torch.Tensor]]:
        """The current ``p.grad`` tensors of the reduced parameters, in buffer order.  A step replayed from a captured
        graph writes its gradients into the tensors that were ``p.grad`` *when that graph was captured*; with one graph
        per resident batch those are different tensors per graph, so the caller keeps this list per graph and hands
        it to ``reduce`` (``p.grad`` itself only names the last captured graph's tensors)."""
        return [p.grad for p in self.params]

    def active_flags(self) -> dict:
        """``{parameter: 0-dim view of its flag}`` for ``FusedAdamW.set_grad_active``: after ``sum_into_flat`` a flag
        of zero means no rank had a gradient for that parameter this step, and the optimiser must leave it alone."""
        return {p: self.flags[i] for i, p in enumerate(self.params)}

    def _fill(self, grads: List[Optional[torch.Tensor]]) -> List[tuple]:
        """Local gradients and has-gradient flags into the buffer; returns the (view, gradient) pairs present.  ONE
        multi-tensor copy: a parameter without a gradient here is filled from a resident zero tensor of its shape (its slot
        holds the other ranks' sum from the previous step otherwise), the flags from the pattern's resident 0 / 1 tensor."""
        have = [(v, g) for v, g in zip(self.views, grads) if g is not None]
        pattern = tuple(g is not None for g in grads)
        local = self._flag_cache.get(pattern)
        if local is None:                             # one small device tensor per None-pattern, built outside any capture
            local = torch.tensor([1.0 if b else 0.0 for b in pattern], dtype=torch.float32).to(self.flags.device)
            if len(self._flag_cache) < 64:
                self._flag_cache[pattern] = local
        src = [g if g is not None else self._zeros_like(i) for i, g in enumerate(grads)]
        if not self._hip_copy(self.views + [self.flags], src + [local]):
            torch._foreach_copy_(self.views + [self.flags], src + [local])
        return have

    def _hip_copy(self, dst: List[torch.Tensor], src: List[torch.Tensor]) -> bool:
        """``mkgnn_flat_copy``: the whole list in one launch (contiguous float32 tensors on one GPU: anything else is left to
        ``torch._foreach_copy_``, two multi-tensor launches of 12 us each where this is one of 6)."""
        if not dst or not dst[0].is_cuda:
            return False
        dev = dst[0].device
        for d, s in zip(dst, src):
            if not (d.is_cuda and s.is_cuda and d.device == dev and s.device == dev and d.dtype == torch.float32
                    and s.dtype == torch.float32 and d.is_contiguous() and s.is_contiguous() and d.numel() == s.numel()):
                return False
        import ctypes as C
        from . import _lib
        key = tuple((d.data_ptr(), s.data_ptr(), d.numel()) for d, s in zip(dst, src) if d.numel())
        table = self._copy_tables.get(key)
        if table is None:                             # (pointers are stable from step to step: one table per gradient pattern)
            table = (_lib.CopyItem * len(key))()
            for e, (dp_, sp_, n) in zip(table, key):
                e.dst, e.src, e.numel = dp_, sp_, n
            if len(self._copy_tables) < 64:
                self._copy_tables[key] = table
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mkgnn_flat_copy(C.cast(table, C.c_void_p), len(key), _lib.stream_ptr(dev)), "mkgnn_flat_copy")
        return True

    def _zeros_like(self, i: int) -> torch.Tensor:
        z = self._zero_cache.get(i)
        if z is None:                                 # (allocated once per parameter that ever lacks a gradient; call
            z = self._zero_cache[i] = torch.zeros_like(self.views[i])      # prepare_patterns before capturing a graph)
        return z

    def prepare_patterns(self, grads_lists) -> None:
        """Build the flag tensors of these gradient lists now (a host-to-device copy cannot happen inside a capture)."""
        for grads in grads_lists:
            pattern = tuple(g is not None for g in grads)
            for i, g in enumerate(grads):
                if g is None:
                    self._zeros_like(i)
            if pattern not in self._flag_cache:
                self._flag_cache[pattern] = torch.tensor([1.0 if b else 0.0 for b in pattern],
                                                         dtype=torch.float32).to(self.flags.device)

    def sum_into_flat(self, grads: List[Optional[torch.Tensor]]) -> None:
        """Copy the given gradients (from ``grads()``) into the flat buffer and SUM it over the ranks; nothing is
        scaled or copied back.  For an optimiser that reads the flat views directly, divides by the world size itself
        and skips the parameters whose flag is zero (``FusedAdamW(grad_scale=1 / world)`` with ``p.grad = view`` and
        ``set_grad_active(active_flags())``): every rank then applies the same update to every parameter, whatever its
        own batch lacked, with two copy launches and one scaling launch fewer per step than ``reduce``."""
        if len(grads) != len(self.params):
            raise ValueError("grads must come from FlatGradAllReduce.grads()")
        self._fill(grads)
        self.all_reduce_filled()

    def fill(self, grads: List[Optional[torch.Tensor]]) -> None:
        """The first half of ``sum_into_flat`` alone -- one multi-tensor copy, capturable: a backward captured as a graph
        ends with it, and the step outside the graph is ``all_reduce_filled()`` only."""
        if len(grads) != len(self.params):
            raise ValueError("grads must come from FlatGradAllReduce.grads()")
        self._fill(grads)

    def all_reduce_filled(self) -> None:
        """SUM the flat buffer (gradient slots and has-gradient flags) over the ranks."""
        if self.world > 1 or dist.is_initialized():          # (a process group of one rank still runs the collective: rehearsals)
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)

    def reduce(self, grads: Optional[List[Optional[torch.Tensor]]] = None) -> None:
        """Call after backward: afterwards every gradient tensor (``p.grad``, or the given list from ``grads()``) holds
        the mean over ranks.  With an explicit list every entry must be a tensor: a ``None`` entry has nowhere to
        receive the other ranks' gradient, and skipping it would let the replicas diverge silently (use
        ``sum_into_flat`` with an optimiser that reads the flat views for that case)."""
        if self.world == 1:
            return
        if grads is not None:
            if len(grads) != len(self.params):
                raise ValueError("grads must come from FlatGradAllReduce.grads()")
            missing = [i for i, g in enumerate(grads) if g is None]
            if missing:
                raise ValueError(f"reduce(grads): {len(missing)} entries are None (first: parameter {missing[0]}); "
                                 "a rank without a gradient tensor cannot receive the reduced one -- use "
                                 "sum_into_flat() with an optimiser that reads the flat views")
            have = self._fill(grads)
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])
            return
        for p in self._filled:                       # gradients this object created last time are not this step's
            p.grad = None
        self._filled = []
        have = self._fill([p.grad for p in self.params])
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / self.world)
        if have:                                     # back into the (possibly graph-static) gradient tensors
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])
        if len(have) != len(self.params):
            # a degree absent from this rank's batch: take the other ranks' mean; absent everywhere: stay None
            # (one small device-to-host copy, only on steps where this rank lacks a gradient)
            counts = self.flags.cpu()
            for i, (v, p) in enumerate(zip(self.views, self.params)):
                if p.grad is None and float(counts[i]) > 0.0:
                    p.grad = v.clone()
                    self._filled.append(p)


class BufferSync:
    """The model's BUFFERS across the ranks: BatchNorm's ``running_mean`` / ``running_var`` are statistics of the rank's own
    batches (``dp.py`` reduces gradients, i.e. parameters, only), so without this rank 0's checkpoint holds rank 0's
    statistics -- where PyTorch's DDP would broadcast rank 0's buffers before every forward.  ``average()`` makes every
    rank's floating-point buffers the MEAN over the ranks (every rank saw 1 / world of the stream: the mean is the
    better estimate of what one process would have accumulated) with ONE all-reduce of one flat buffer, and takes the
    integer buffers (``num_batches_tracked``) from rank 0; ``broadcast()`` is DDP's form (everything from rank 0).  Call
    it at checkpoints / ends of epochs, outside captured steps; ``max_abs_diff()`` says how far the replicas' buffers are
    from rank 0's (0 after either call)."""

    def __init__(self, module: torch.nn.Module):
        named = [(n, b) for n, b in module.named_buffers() if b is not None and b.numel()]
        self.float_names = [n for n, b in named if b.dtype.is_floating_point]
        self.floats = [b for _, b in named if b.dtype.is_floating_point]
        self.others = [b for _, b in named if not b.dtype.is_floating_point]
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def _flat(self) -> Optional[torch.Tensor]:
        if not self.floats:
            return None
        return torch.cat([b.detach().reshape(-1).to(torch.float32) for b in self.floats])

    def _scatter_back(self, flat: torch.Tensor) -> None:
        off = 0
        with torch.no_grad():
            for b in self.floats:
                b.copy_(flat[off:off + b.numel()].view_as(b).to(b.dtype))
                off += b.numel()

    def average(self) -> None:
        if self.world == 1 and not dist.is_initialized():
            return
        flat = self._flat()
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.mul_(1.0 / self.world)
            self._scatter_back(flat)
        for b in self.others:
            dist.broadcast(b, 0)

    def broadcast(self) -> None:
        if self.world == 1 and not dist.is_initialized():
            return
        flat = self._flat()
        if flat is not None:
            dist.broadcast(flat, 0)
            self._scatter_back(flat)
        for b in self.others:
            dist.broadcast(b, 0)

    def max_abs_diff(self) -> float:
        """max over ranks and buffers of |buffer - rank 0's| (a collective: every rank calls it)."""
        if self.world == 1 and not dist.is_initialized():
            return 0.0
        parts = [b.detach().reshape(-1).to(torch.float64) for b in self.floats + self.others]
        if not parts:
            return 0.0
        mine = torch.cat(parts)
        ref = mine.clone()
        dist.broadcast(ref, 0)
        d = (mine - ref).abs().max().reshape(1)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        return float(d.item())


# parameter-name fragments that never receive a gradient in the reference's model; "^" anchors a
# prefix: GNNModel's own unused heads are lin1 / lin2 (model.py:147-148), while gnn_model.graph_embedding_lin1 / lin2
# ARE trained and must be reduced
NEVER_TRAINED = ("p_support", "length_sc_weight", "angle_sc_weight", "edge_batch_norm", "graph_embedding_linear",
                 "^lin1.", "^lin2.")
