"""Cycle totals of the fused tail's middle kernel per phase (thread 0 of every workgroup): tools/tail_stamps.py [--batch-size N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib, readout as R            # noqa: E402
from molkgnn_amd.plan import plan_from_data          # noqa: E402
from molkgnn_amd.synthetic import make_batch         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-size", type=int, default=4096)
args = ap.parse_args()
dev = torch.device("cuda:0")
import ctypes                                        # noqa: E402
lib = _lib.load()
lib.mkgnn_debug_set_tail_stamps.restype = ctypes.c_int
lib.mkgnn_debug_set_tail_stamps.argtypes = [ctypes.c_void_p]      # (a 64-bit device pointer: never through the default int)
b = make_batch(args.batch_size, seed=1798000).to(dev)
plan = plan_from_data(b)
seg = R.molecule_segments(b.batch, args.batch_size)
Ls = (10, 20, 30, 50)
torch.manual_seed(0)
lin1, lin2, ffn = torch.nn.Linear(110, 32).to(dev), torch.nn.Linear(32, 32).to(dev), torch.nn.Linear(32, 1).to(dev)
sim = torch.randn(b.x.shape[0], 112, device=dev)[:, :110].requires_grad_(True)
y = torch.zeros(args.batch_size, device=dev)
buf = torch.zeros(1024 * 16, dtype=torch.int64, device=dev)      # (>= 512 workgroups x 16 words)
for it in range(3):
    lib.mkgnn_debug_set_tail_stamps(buf.data_ptr() if it == 2 else None)
    loss = R.tail_loss(sim, plan, Ls, lin1, lin2, ffn, y, seg, 0.25, None)
    torch.cuda.synchronize()
lib.mkgnn_debug_set_tail_stamps(None)
st = buf.view(1024, 16).cpu().double()
st = st[st[:, :9].sum(dim=1) > 0]
names = ["prologue", "window", "chunk loads", "P3 propagate+swish", "P4 molecules", "dW2+P5", "P6 propagate^T", "final sync", "epilogue"]
tot = st[:, :9].sum(dim=1)
print(f"{st.shape[0]} workgroups; total cycles per workgroup: mean {tot.mean():.0f} max {tot.max():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:22s} mean {st[:, i].mean():9.0f}  max {st[:, i].max():9.0f}")
