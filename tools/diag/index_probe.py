"""Kernels of the index builders on one batch (for rocprofv3 --kernel-trace --stats): the one-pass builder, then the
separate receptive-field + plan builders."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd.plan import plan_from_lists                     # noqa: E402
from molkgnn_amd.receptive_field import build_index_hip, build_receptive_fields_hip      # noqa: E402
from molkgnn_amd.synthetic import make_batch                     # noqa: E402

dev = torch.device("cuda:0")
b = make_batch(4096, seed=5, device=dev, with_receptive_fields=False)
f = build_receptive_fields_hip(b.x, b.p, b.edge_index, b.edge_attr)
sizes = [int(f[f"selected_index_deg{d}"].numel()) for d in range(1, 5)]
os.environ["MKGNN_INDEX_OVERLAP"] = "0"
for _ in range(20):
    build_index_hip(b.x, b.p, b.edge_index, b.edge_attr, sizes)
torch.cuda.synchronize()
for _ in range(20):
    rf = build_receptive_fields_hip(b.x, b.p, b.edge_index, b.edge_attr, sizes)
    lists = [[rf[f"{nm}_deg{d}"] for d in range(1, 5)] for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index")]
    plan_from_lists(b.x.shape[0], *lists, b.edge_index).build_hip()
torch.cuda.synchronize()
print("done")
