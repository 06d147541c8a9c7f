"""Forward of one N-hop layer with and without the saved backward state (ids, scores): run under
rocprofv3 --kernel-trace and compare the two halves of the kc_forward_fused calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

dev = torch.device("cuda:0")
b = make_batch(4096, seed=1798000).to(dev)
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=110, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
params = [p.detach() for p in params]
store = torch.zeros(b.x.shape[0], 112, device=dev)
store[:, :110] = torch.rand(b.x.shape[0], 110, device=dev) * 2 - 1
x = store[:, :110]
with torch.no_grad():
    for saved in (True, False):
        for _ in range(12):
            Fn._forward_impl(x, plan, False, Fn.VARIANTS["auto"], None, E, params, saved)
torch.cuda.synchronize()
print("done")
