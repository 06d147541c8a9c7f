"""One small resident batch (default: AID 435008 shape, 256 molecules) through the training step as a replayed hipGraph:
ms per step; under rocprofv3 --kernel-trace --stats the per-kernel durations of the molecule-resident path
(MKGNN_MOLECULE=0: the per-operator path).   python3 tools/small_batch_probe.py [batch] [replays]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd.synthetic import make_batch                                   # noqa: E402
from molkgnn_amd.train import GNNModel, configure_optimizer                    # noqa: E402
from molkgnn_amd.train import backward as train_backward                       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
torch.manual_seed(1798)
model = GNNModel(ffn_dropout_rate=0.25).to(dev).train()
opt = configure_optimizer(model, lr=1e-3, capturable=True)
batches = [make_batch(B, seed=435008000 + 700 + i, assay="435008").to(dev) for i in range(4)]
graphs = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for b in batches:
        model.zero_grad(set_to_none=True)
        train_backward(model.loss(b))
        opt.step()
    for b in batches:
        model.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            train_backward(model.loss(b))
            opt.step()
        graphs.append(g)
torch.cuda.current_stream().wait_stream(side)
for g in graphs:
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(reps):
    graphs[i % 4].replay()
torch.cuda.synchronize()
print(f"batch {B}: {1e3 * (time.perf_counter() - t0) / reps:.4f} ms per step ({batches[0].x.shape[0]} atoms)")
