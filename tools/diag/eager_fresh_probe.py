"""Eager training steps on FRESH batches (every step a batch the process has not seen: plans, chunk tables and caches are
built inside the timed loop), the default dispatch against MKGNN_MOLECULE=0 (is the molecule-resident step
still the faster eager path when nothing is cached?).  tools/diag/eager_fresh_probe.py [--molecules 256] [--steps 96]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd import molecule as M                     # noqa: E402
from molkgnn_amd.synthetic import make_batch              # noqa: E402
from molkgnn_amd.train import GNNModel, configure_optimizer            # noqa: E402
from molkgnn_amd.train import backward as train_backward  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--molecules", type=int, default=256)
ap.add_argument("--steps", type=int, default=96)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for mode in ("", "0"):
    M._MODE = mode
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    opt = configure_optimizer(model, lr=1e-3, fused=True)
    host = [make_batch(a.molecules, seed=9000 + i) for i in range(a.steps + 8)]      # (built on the host before the clock starts)
    for b in host:
        b.num_graphs = a.molecules

    def step(b):
        model.zero_grad(set_to_none=True)
        train_backward(model.loss(b))
        opt.step()

    for b in host[:8]:
        step(b.to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in host[8:]:
        step(b.to(dev))                                   # host-to-device copy of a fresh batch, its plan, the step
    torch.cuda.synchronize()
    fresh = (time.perf_counter() - t0) / a.steps
    res = [b.to(dev) for b in host[8:12]]
    for b in res:
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(res[i % 4])
    torch.cuda.synchronize()
    resident = (time.perf_counter() - t0) / a.steps
    print(f"{a.molecules} molecules, MKGNN_MOLECULE={'unset (one-launch step)' if mode == '' else '0 (per operator)'}: "
          f"{1e3 * fresh:.3f} ms per eager step on fresh batches, {1e3 * resident:.3f} ms on resident ones")
