"""The five kernels of one N-hop layer's backward (batch 4096, F = K = 110), each alone on the GPU: HIP-event durations
(mkgnn_debug_time_backward).  MKGNN_LIB=<variant build> for A/B runs.  For counters:
    tools/pmc.sh <out dir> "<counters>" -- tools/bwd_kernel_times.py --plain
(pmc.sh puts `python3 <script>` itself behind rocprofv3's `--`; never run this file through an env shebang or a shell
wrapper under the profiler: its preloaded library has initialised the GPU by then, and an exec from there is refused
on this pool.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                         # noqa: E402
from molkgnn_amd import _lib                         # noqa: E402
from molkgnn_amd import functional as Fn             # noqa: E402
from molkgnn_amd.kernels import KernelSetConv        # noqa: E402
from molkgnn_amd.plan import plan_from_data          # noqa: E402
from molkgnn_amd.synthetic import make_batch         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-size", type=int, default=4096)
ap.add_argument("--width", type=int, default=110)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--plain", action="store_true", help="no event timing: just run the backward `reps` times (for rocprofv3)")
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
h = store[:, :F]
lib = _lib.load()
if args.plain:
    x = h.detach().requires_grad_(True)
    wgt = torch.randn(h.shape[0], 110, device=dev)
    for _ in range(args.reps):
        for p in params:
            p.grad = None
        x.grad = None
        (Fn.kernelsetconv(x, plan, False, params, E, "auto", block_rows=True, propagate=True) * wgt).sum().backward()
    torch.cuda.synchronize()
    sys.exit(0)
t = bench.time_backward_kernels(lib, Fn, h, plan, params, E, "auto", args.reps)
alg = bench.backward_algorithmic(plan, F, E, [10, 20, 30, 50])
for name, ms in t.items():
    by, fl = alg[name]
    print(f"{name:28s} {1e3 * ms:7.1f} us   {by / ms / 1e6:7.0f} GB/s algorithmic   {fl / ms / 1e9:6.1f} TFLOP/s useful")
print(f"{'sum':28s} {1e3 * sum(t.values()):7.1f} us")
