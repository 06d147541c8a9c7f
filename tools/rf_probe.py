"""Time the receptive-field builders on a resident batch: torch index arithmetic vs the HIP passes (f-2)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd.receptive_field import build_receptive_fields, build_receptive_fields_hip   # noqa: E402
from molkgnn_amd.synthetic import make_batch                                                  # noqa: E402

dev = torch.device("cuda:0")
nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = make_batch(nmol, seed=1798000, with_receptive_fields=False).to(dev)
for name, fn in (("torch", build_receptive_fields), ("hip", build_receptive_fields_hip)):
    for _ in range(3):
        fn(b.x, b.p, b.edge_index, b.edge_attr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        fn(b.x, b.p, b.edge_index, b.edge_attr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:5s}: {dt * 1e3:7.3f} ms per batch of {nmol} molecules ({b.x.shape[0]} atoms, {b.edge_index.shape[1]} edges) "
          f"= {nmol / dt / 1e6:.2f} M molecules/s")
