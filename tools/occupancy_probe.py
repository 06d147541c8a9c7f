"""Forward kernel on a batch whose atoms all have ONE degree (rings: 2; bond pairs: 1), for occupancy experiments with
single-degree builds (make VARIANT=... EXTRA="-DMKGNN_EXP_ONLY_D=2 -DMKGNN_EXP_OCC=3 -DMKGNN_EXP_MAX_BLOCKS=768"):
tools/occupancy_probe.py <degree 1|2> [atoms]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib                                            # noqa: E402
from molkgnn_amd import functional as Fn                                # noqa: E402
from molkgnn_amd.kernels import KernelSetConv                           # noqa: E402
from molkgnn_amd.plan import plan_from_data                             # noqa: E402
from molkgnn_amd.receptive_field import GraphBatch, attach_receptive_fields   # noqa: E402

deg = int(sys.argv[1])
n_atoms = int(sys.argv[2]) if len(sys.argv) > 2 else 102400
dev = torch.device("cuda:0")
torch.manual_seed(0)
size = 2 if deg == 1 else 25
m = n_atoms // size
base = torch.arange(m).repeat_interleave(size) * size
k = torch.arange(size).repeat(m)
if deg == 1:
    i = torch.arange(m) * 2
    bonds = torch.stack([i, i + 1], 1)
else:
    bonds = torch.stack([base + k, base + (k + 1) % size], 1)
ei = torch.stack([torch.stack([bonds[:, 0], bonds[:, 1]], 1).reshape(-1), torch.stack([bonds[:, 1], bonds[:, 0]], 1).reshape(-1)])
nb = bonds.shape[0]
ea = torch.zeros(nb, 7); ea[torch.arange(nb), torch.randint(0, 4, (nb,))] = 1.0
b = GraphBatch(x=torch.randn(m * size, 28), p=torch.randn(m * size, 3), edge_index=ei, edge_attr=ea.repeat_interleave(2, 0),
               batch=torch.arange(m).repeat_interleave(size), y=torch.zeros(m), num_graphs=m).to(dev)
attach_receptive_fields(b)
plan = plan_from_data(b)
print("bucket sizes", [bk.count for bk in plan.buckets])
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=110, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
store = torch.zeros(b.x.shape[0], 112, device=dev)
store[:, :110] = torch.rand(b.x.shape[0], 110, device=dev) * 2 - 1
x = store[:, :110]
lib = _lib.load()
lib.mkgnn_debug_last_fused_forward_ms.restype = ctypes.c_float
for _ in range(5):
    Fn.kernelsetconv_details(x, plan, False, params, E, "auto", raw=True)
lib.mkgnn_debug_time_fused_forward(1)
ts = []
for _ in range(30):
    out, _ = Fn.kernelsetconv_details(x, plan, False, params, E, "auto", raw=True)
    ts.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
lib.mkgnn_debug_time_fused_forward(0)
ts.sort()
print(f"degree {deg}: forward launch us: min {1e3 * ts[0]:.1f} median {1e3 * ts[15]:.1f}; checksum {float(out.double().sum()):.6f}")
