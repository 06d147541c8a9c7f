"""Run only the kernel-convolution forward (and optionally backward) of one layer, for rocprofv3 runs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-size", type=int, default=4096)
ap.add_argument("--width", type=int, default=110)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--variant", default="auto")
ap.add_argument("--backward", action="store_true")
ap.add_argument("--last", action="store_true")
ap.add_argument("--fp32-rows", action="store_true", help="feed fp32 rows (kc_forward_stream<KC, 2>) instead of the pre-split rows the "
                "training step feeds (kc_forward_stream<KC, 3>)")
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
if not args.fp32_rows and args.variant in ("auto", "mfma") and Fn.rows_split_supported(plan, params, F, E, plan.n_atoms):
    xs = Fn.presplit_rows(x)                       # the operand form of the step (functional.ROWS_SPLIT)
    x = xs.detach()
    setattr(x, Fn._INV_ATTR, (getattr(xs, Fn._INV_ATTR)[0], x._version))
    Fn.mark_rows_split(x)
x = x.requires_grad_(args.backward)
for _ in range(args.reps):
    if args.backward:
        out = Fn.kernelsetconv(x, plan, args.last, params, E, args.variant)
        out.backward(torch.ones_like(out))
    else:
        Fn.kernelsetconv_details(x, plan, args.last, params, E, args.variant)
torch.cuda.synchronize()
print("done", b.x.shape[0], "atoms", [bk.count for bk in plan.buckets])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import layer_algorithmic                 # noqa: E402
print("algorithmic_bytes=%d algorithmic_flops=%d" % layer_algorithmic(plan, F, E, layer.L, args.last))
if not args.backward and args.variant != "generic":
    # the fused forward launch alone: HIP events recorded around it on its own stream (several rounds, sorted)
    import ctypes
    from molkgnn_amd import _lib
    lib = _lib.load()
    lib.mkgnn_debug_last_fused_forward_ms.restype = ctypes.c_float
    lib.mkgnn_debug_time_fused_forward(1)
    ms = []
    for _ in range(max(args.reps, 10)):
        Fn.kernelsetconv_details(x, plan, args.last, params, E, args.variant)
        ms.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
    lib.mkgnn_debug_time_fused_forward(0)
    ms.sort()
    print("fused forward launch us: min %.1f median %.1f max %.1f" % (1e3 * ms[0], 1e3 * ms[len(ms) // 2], 1e3 * ms[-1]))
