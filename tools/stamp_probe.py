"""Diagnostic: cycle stamps inside the fused forward kernel, grouped by (degree, column part).
Needs a library built with the stamps compiled in: make -C molkgnn_amd/csrc clean && make -C molkgnn_amd/csrc STAMPS=1
(they cost ~5 us of the kernel and perturb its wait-count placement, so the default build leaves them out)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib                         # noqa: E402
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
b = make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, seed=1798000).to(dev)
VARIANT = sys.argv[2] if len(sys.argv) > 2 else "mfma"        # "mfma" (fp32) or "bf16"
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=110, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
store = torch.zeros(b.x.shape[0], 112, device=dev)
store[:, :110] = torch.rand(b.x.shape[0], 110, device=dev) * 2 - 1
x = store[:, :110]
for _ in range(3):
    Fn.kernelsetconv_details(x, plan, False, params, E, VARIANT)
buf = torch.zeros(4096 * 32, dtype=torch.int64, device=dev)
assert lib.mkgnn_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
Fn.kernelsetconv_details(x, plan, False, params, E, VARIANT)
torch.cuda.synchronize()
lib.mkgnn_debug_set_stamp_buffer(C.c_void_p(0))
st = buf.cpu().numpy().reshape(-1, 32)
st = st[st[:, 0] != 0]
t0 = st[:, 0].min()
print(f"waves {st.shape[0]}; start skew p50 {np.median(st[:,0]-t0):.0f} max {(st[:,0]-t0).max()}")
last = np.array([row[2:29][row[2:29] != 0].max() if (row[2:29] != 0).any() else row[1] for row in st])
print(f"kernel span (first start -> last stamp): {last.max() - t0} cycles")
for key in sorted(set(st[:, 31])):
    w = st[st[:, 31] == key]
    nun = ((w[:, 2:20] != 0).sum(1) // 3)
    bank = w[:, 1] - w[:, 0]
    mult, store_, per = [], [], []
    ends = []
    for r in range(w.shape[0]):
        for u in range(nun[r]):
            a0, a1, a2 = w[r, 2 + 3 * u: 5 + 3 * u]
            mult.append(a1 - a0); store_.append(a2 - a1)
        ends.append((w[r, 2 + 3 * nun[r] - 1] if nun[r] else w[r, 1]) - t0)
    print(f"deg {key // 16} cp {key % 16}: waves {w.shape[0]:4d} tiles/wave {nun.mean():.2f} (max {nun.max()}) | bank {np.median(bank):6.0f} "
          f"| multiply med {np.median(mult):6.0f} p90 {np.percentile(mult, 90):6.0f} | epilogue med {np.median(store_):5.0f} "
          f"| first tile: perm {np.median(w[:,20]-w[:,3]):6.0f} bonds {np.median(w[:,21]-w[:,20]):6.0f} rest {np.median(w[:,4]-w[:,21]):6.0f}")
