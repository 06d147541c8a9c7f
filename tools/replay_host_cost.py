"""Host time of one hipGraph replay of the training step against its GPU time (is the replay loop host- or GPU-bound?)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd.synthetic import make_batch                                   # noqa: E402
from molkgnn_amd.train import GNNModel, backward, configure_optimizer          # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = make_batch(B, seed=1).to(dev)
model = GNNModel(num_layers=3).to(dev)
opt = configure_optimizer(model, lr=1e-3, capturable=True)


def step():
    model.zero_grad(set_to_none=True)
    loss = model.loss(b)
    backward(loss)
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
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# host cost with an idle GPU queue: replay, wait, replay, ...
hs = []
for _ in range(50):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    g.replay()
    hs.append(time.perf_counter() - t1)
hs.sort()
print(f"batch {B}: {N} replays: host loop {1e3 * t_host / N:.3f} ms per replay (queue allowed to fill), all done after {1e3 * t_all / N:.3f} ms "
      f"per replay; replay() on an idle queue: median {1e3 * hs[25]:.3f} ms")
