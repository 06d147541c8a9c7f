"""Diagnostic: which Python lines launch the small PyTorch kernels (fills, adds, copies) of one training step.
usage: tools/op_trace.py [batch]   -- eager step under torch.profiler with stacks; prints op, shape and the nearest frames."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib                                  # noqa: E402
from molkgnn_amd.synthetic import make_batch                  # noqa: E402
from molkgnn_amd.train import GNNModel, configure_optimizer   # noqa: E402

_lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GNNModel().to(dev).train()
opt = configure_optimizer(model, capturable=True)
b = make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 16, seed=1).to(dev)


def step():
    model.zero_grad(set_to_none=True)
    loss = model.loss(b)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::copy_", "aten::mul", "aten::ones_like", "aten::zeros", "aten::clone",
        "aten::_foreach", "aten::_fused", "aten::sum", "aten::native_dropout", "aten::cat", "aten::index")
for ev in sorted(prof.events(), key=lambda e: e.time_range.start):
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith(want):
        continue
    if not any(k.device_time_total > 0 for k in [ev]) and ev.device_time_total == 0:
        continue
    frames = [f for f in (ev.stack or []) if "molkgnn_amd" in f or "bench" in f or "optim" in f or "op_trace" in f][:3]
    print(f"{ev.name:28s} {str(ev.input_shapes)[:60]:60s} dev {ev.device_time_total:6.1f} us | {' <- '.join(frames)[:200]}")
