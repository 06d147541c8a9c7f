"""Diagnostic: the three propagate passes (dense, block-row sources, block-row destinations) alone, for rocprofv3
kernel traces / PMC passes: tools/csr_probe.py [batch] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

dev = torch.device("cuda:0")
b = make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, seed=1798000).to(dev)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
plan = plan_from_data(b)
Ls = (10, 20, 30, 50)
n, K = b.x.shape[0], sum(Ls)
store = torch.rand(n, K + (-K) % 4, device=dev)
v = store[:, :K]
vb = store[:, :K]
setattr(vb, Fn._BLOCKS_ATTR, Ls)
for _ in range(reps):
    Fn.propagate_add(store[:, :K], plan, out_pad=(-K) % 4)                       # dense
    Fn.propagate_add(vb, plan, out_pad=(-K) % 4)                                 # block-row sources
    Fn._segment_sum_blocks(v, plan.csr_out, plan.deg8, Ls, 2, (-K) % 4, None)    # block-row destinations
torch.cuda.synchronize()
print("done")
