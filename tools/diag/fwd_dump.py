"""Dump / compare the streamed forward's output on one seeded batch (diagnostics: A/B of kernel forms selected by
environment variables read at library load).  fwd_dump.py save|cmp FILE [--width W] [--batch-size B] [--last]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("mode", choices=("save", "cmp"))
ap.add_argument("file")
ap.add_argument("--batch-size", type=int, default=512)
ap.add_argument("--width", type=int, default=110)
ap.add_argument("--last", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
b = make_batch(args.batch_size, seed=1798000).to(dev)
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=args.width, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
F = args.width
store = torch.zeros(b.x.shape[0], F + (-F) % 4, device=dev)
store[:, :F] = torch.rand(b.x.shape[0], F, device=dev) * 2 - 1
x = store[:, :F]
out, saved = Fn.kernelsetconv_details(x, plan, args.last, params, E, "auto")
torch.cuda.synchronize()
import ctypes                                       # noqa: E402
from molkgnn_amd import _lib                        # noqa: E402
plans = (ctypes.c_int32 * 12)()
if hasattr(_lib.load(), "mkgnn_debug_last_plans") and _lib.load().mkgnn_debug_last_plans(plans) == 0:
    print("forward plan: blocks", plans[0], "launches", plans[9])
out = out.cpu()
if args.mode == "save":
    torch.save({"out": out, "saved": [None if s[0] is None else s[0].cpu() for s in saved]}, args.file)
    print("saved", tuple(out.shape))
else:
    ref = torch.load(args.file)
    d = (out - ref["out"]).abs()
    print("max abs diff", float(d.max()), "rows differing", int((d.max(dim=1).values > 0).sum()), "of", out.shape[0])
    off = 0
    for i, bk in enumerate(plan.buckets):
        L = (10, 20, 30, 50)[i]
        if bk.count:
            sel = bk.sel.cpu()
            dd = d[sel][:, off:off + L]
            bad = (dd.max(dim=1).values > 0).nonzero().reshape(-1)
            print(f"degree {i + 1}: {bk.count} atoms, {bad.numel()} rows differ; first bad bucket positions {bad[:12].tolist()}; "
                  f"bad per 16-tile {torch.bincount(bad // 16)[:24].tolist() if bad.numel() else []}")
        off += L
