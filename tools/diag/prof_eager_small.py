"""cProfile of the eager training step at a small batch (host side): tools/diag/prof_eager_small.py [batch]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd import molecule as Mol                                       # noqa: E402
from molkgnn_amd.synthetic import make_batch                                  # noqa: E402
from molkgnn_amd.train import GNNModel, backward, configure_optimizer        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GNNModel().to(dev).train()
opt = configure_optimizer(model, lr=1e-3, capturable=True)
Mol._MODE = "0"
batches = [make_batch(B, seed=900 + i, assay="435008").to(dev) for i in range(4)]


def step(i):
    model.zero_grad(set_to_none=True)
    backward(model.loss(batches[i % 4]))
    opt.step()


for i in range(10):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    step(i)
torch.cuda.synchronize()
print(f"eager step: {(time.perf_counter() - t0) * 10:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(100):
    step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
