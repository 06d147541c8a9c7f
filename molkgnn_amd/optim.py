"""AdamW for the MolKGNN training step, one HIP launch for the whole model (``mkgnn_adamw_step``).

The reference builds ``torch.optim.AdamW`` over two parameter groups (``model.py:368-385``).  This class keeps that
interface -- ``param_groups`` with ``lr`` / ``betas`` / ``eps`` / ``weight_decay`` / ``maximize``, per-parameter state
``step`` / ``exp_avg`` / ``exp_avg_sq``, ``state_dict`` round trips, parameters without a gradient skipped -- and the
same update formula; the ~80 small tensors of the model are updated by one kernel instead of PyTorch's five
(``csrc/kgnn_optim.hip``).  The step counters live on the device, so a step can be captured in a hipGraph; a
learning-rate schedule reaches a captured step through ``lr`` given as a 0-dim CUDA tensor that the scheduler fills.
fp32 CUDA parameters only; no CPU path (``MolKGNNLibraryError`` otherwise).
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, maximize: bool = False,
                 grad_scale: float = 1.0):
        if not isinstance(lr, torch.Tensor) and lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid betas: {betas}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if weight_decay < 0.0:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        # grad_scale: every gradient is multiplied by it first -- 1 / world after a summing all-reduce, so that the
        # averaging costs no kernel of its own (default 1: torch.optim.AdamW's update)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, maximize=maximize,
                                      grad_scale=grad_scale))
        if len(self.param_groups) > 4:
            raise ValueError("FusedAdamW takes at most 4 parameter groups")
        self._table_key = None
        self._table = None
        self._active: dict = {}
        # per-parameter constants of step() -- group, packed-state pointer, element count, active-flag pointer -- looked up once
        # and kept while nothing that determines them changes (an eager step of this model is host-bound: ~80 tensors)
        self._fast: list = []
        self._fast_sig = None
        self._ver = 0

    def set_grad_active(self, flags: dict) -> None:
        """``{parameter: one-element float32 CUDA tensor}``: a parameter whose flag reads 0 when the step runs is
        skipped (as if its gradient were ``None``) -- decided on the device, so a captured data-parallel step skips the
        banks of a degree that no rank's batch contained (``dp.FlatGradAllReduce.active_flags``)."""
        for p, f in flags.items():
            if not (isinstance(f, torch.Tensor) and f.is_cuda and f.dtype == torch.float32 and f.numel() == 1):
                raise ValueError("an active flag must be one float32 element on the GPU")
        self._active = dict(flags)
        self._table_key = None
        self._ver += 1
        self.prepare_state(list(flags))                      # flagged parameters are stepped on the device's say-so: state must exist

    # -- state: one buffer per parameter, [exp_avg | exp_avg_sq | step | 2 reserved | a step count per 1024 elements]
    # (mkgnn_adamw_state_floats); the three state entries are views of it --
    @staticmethod
    def _state_floats(n: int) -> int:
        return 2 * n + 3 + (n + 1023) // 1024

    def _packed_state(self, p: torch.Tensor) -> torch.Tensor:
        st = self.state[p]
        n = p.numel()
        buf = st.get("_packed")
        if buf is not None and st["exp_avg"].data_ptr() == buf.data_ptr() and st["step"].data_ptr() == buf[2 * n:].data_ptr() and buf.numel() == self._state_floats(n):
            return buf
        if p.is_cuda and torch.cuda.is_current_stream_capturing():
            # a zero fill recorded into a graph would reset exp_avg / exp_avg_sq / step at EVERY replay (and, data-parallel,
            # on this rank only: the replicas would drift apart silently)
            raise _lib.MolKGNNLibraryError(
                "FusedAdamW: optimiser state would be allocated inside a hipGraph capture -- call prepare_state() (or run "
                "one eager step in which every parameter has a gradient) before capturing")
        new = torch.zeros(self._state_floats(n), dtype=torch.float32, device=p.device)
        if "exp_avg" in st:                                  # e.g. loaded from a state_dict (ours or torch.optim.AdamW's)
            new[:n] = st["exp_avg"].reshape(-1).to(new)
            new[n:2 * n] = st["exp_avg_sq"].reshape(-1).to(new)
            new[2 * n] = float(st["step"])
            new[2 * n + 3:] = float(st["step"])              # (the update's per-block copies of the step count)
        st["_packed"] = new
        st["exp_avg"] = new[:n].view_as(p)
        st["exp_avg_sq"] = new[n:2 * n].view_as(p)
        st["step"] = new[2 * n:2 * n + 1].view(())
        self._table_key = None
        self._ver += 1
        return new

    @torch.no_grad()
    def prepare_state(self, params=None) -> None:
        """Materialise the packed state of ``params`` (default: every parameter of every group) now, outside any capture:
        a parameter whose gradient is absent in the eager warm-up steps -- a degree bank that this rank's warm-up batches do
        not contain -- must not get its zero-filled state as a node of a captured graph (``_packed_state`` refuses)."""
        for p in (params if params is not None else [q for g in self.param_groups for q in g["params"]]):
            if p.numel():
                _lib.require_gpu_tensor(p, "parameter")
                self._packed_state(p)

    def state_dict(self):
        sd = super().state_dict()                            # the packed buffer is an implementation detail: its three views are saved
        sd["state"] = {k: {kk: vv for kk, vv in v.items() if kk != "_packed"} for k, v in sd["state"].items()}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._ver += 1
        for p in list(self.state):                           # pack now: torch may hand over the caller's tensors uncopied
            if isinstance(p, torch.Tensor) and "exp_avg" in self.state[p]:
                self.state[p].pop("_packed", None)
                self._packed_state(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        rows: List[tuple] = []
        keep = []                                            # tensors the enqueued kernel reads: referenced until the call returns
        dev = None
        sig = (self._ver, tuple(len(g["params"]) for g in self.param_groups))
        if sig != self._fast_sig:                            # (param_groups, state or flags changed: look everything up again)
            self._fast = [[p, gi, None, p.numel(), None] for gi, group in enumerate(self.param_groups) for p in group["params"] if p.numel()]
            self._fast_sig = sig
        for ent in self._fast:
            p = ent[0]
            g = p.grad
            if g is None:
                continue
            if ent[2] is None or ent[2][1] != p.data_ptr():  # first step with a gradient (or the parameter's storage moved)
                _lib.require_gpu_tensor(p, "parameter")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.MolKGNNLibraryError("FusedAdamW needs contiguous float32 parameters")
                ver = self._ver
                st_ptr = self._packed_state(p).data_ptr()    # (may allocate: bumps _ver, the list is rebuilt next step)
                act = self._active.get(p)
                ent[2], ent[4] = (st_ptr, p.data_ptr()), (None if act is None else act.data_ptr())
                if ver != self._ver:
                    self._fast_sig = None
            else:
                # a state entry replaced -- or the whole state cleared -- from outside (not through load_state_dict)
                ea = self.state[p].get("exp_avg")
                if ea is None or ea.data_ptr() != ent[2][0]:
                    ent[2] = (self._packed_state(p).data_ptr(), p.data_ptr())
            if g.is_sparse:
                raise RuntimeError("FusedAdamW does not support sparse gradients")
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
                keep.append(g)
            if dev is None:
                dev = p.device
            elif p.device != dev:
                raise _lib.MolKGNNLibraryError("FusedAdamW: parameters on more than one device")
            rows.append((ent[2][1], g.data_ptr(), ent[2][0], ent[3], ent[1], ent[4]))
        if not rows:
            return loss
        key = tuple(rows)
        if key != self._table_key:                           # pointers are stable from step to step: build the table once
            table = (_lib.AdamWTensor * len(rows))()
            for e, r in zip(table, rows):
                e.param, e.grad, e.state, e.numel, e.group, e.active = r
            self._table, self._table_key = table, key
        groups = (_lib.AdamWGroup * len(self.param_groups))()
        for e, group in zip(groups, self.param_groups):
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                if lr.is_cuda:
                    if lr.dtype != torch.float32 or lr.numel() != 1:
                        raise ValueError("a tensor lr must be one float32 element")
                    e.lr_device, e.lr = lr.data_ptr(), 0.0
                else:
                    e.lr_device, e.lr = None, float(lr)
            else:
                e.lr_device, e.lr = None, float(lr)
            e.beta1, e.beta2 = float(group["betas"][0]), float(group["betas"][1])
            e.eps, e.weight_decay, e.maximize = float(group["eps"]), float(group["weight_decay"]), int(bool(group["maximize"]))
            e.grad_scale = float(group.get("grad_scale", 1.0))
        with torch.cuda.device(dev):
            _lib.check(lib.mkgnn_adamw_step(self._table, len(rows), groups, len(groups), _lib.stream_ptr(dev)),
                       "mkgnn_adamw_step")
        del keep
        return loss
