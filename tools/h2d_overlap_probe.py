"""Do host-to-device copies overlap a replayed training-step graph on this stack?  Times N iterations of (a) the graph alone,
(b) a 14.6 MB pinned copy alone on a second stream, (c) both issued together with no dependence between them."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd.synthetic import make_batch                                   # noqa: E402
from molkgnn_amd.train import GNNModel, backward, configure_optimizer          # noqa: E402

dev = torch.device("cuda:0")
b = make_batch(4096, seed=1).to(dev)
model = GNNModel(num_layers=3).to(dev)
opt = configure_optimizer(model, lr=1e-3, capturable=True)


def step():
    model.zero_grad(set_to_none=True)
    backward(model.loss(b))
    opt.step()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
torch.cuda.current_stream().wait_stream(side)
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 14_600_000
pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
dst = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(4)]
copy_stream = torch.cuda.Stream()
N = 200


def timed(graph, copy):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        if copy:
            with torch.cuda.stream(copy_stream):
                dst[i & 3].copy_(pin, non_blocking=True)
        if graph:
            g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


for _ in range(2):
    a, c, both = timed(True, False), timed(False, True), timed(True, True)
print(f"graph alone {a:.3f} ms, copy of {nbytes / 1e6:.1f} MB alone {c:.3f} ms ({nbytes / c / 1e6:.1f} GB/s), together {both:.3f} ms per iteration "
      f"(perfect overlap {max(a, c):.3f}, none {a + c:.3f})")
