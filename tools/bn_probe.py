"""Device time of the batch norm's launches alone (a hipGraph of REPS calls, replayed): forward with / without the statistics
companion, backward.  tools/bn_probe.py [--atoms N] [--bonds M]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import readout as R            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--atoms", type=int, default=102584)
ap.add_argument("--bonds", type=int, default=215920)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(args.atoms, 28, device=dev, requires_grad=True)
e = torch.randn(args.bonds, 7, device=dev)
key = torch.randint(0, args.atoms, (args.bonds,), device=dev)
lim = torch.tensor([args.atoms], device=dev)
bn = torch.nn.BatchNorm1d(28).to(dev).train()
bn2 = torch.nn.BatchNorm1d(7).to(dev).train()
g_out = torch.randn(args.atoms, 28, device=dev)


def timed(fn, label):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(args.reps):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 / args.reps)
    print(f"{label:40s} {best * 1e6:7.2f} us per call")


with torch.no_grad():
    timed(lambda: R.batch_norm(x, bn), "forward, no companion")
    timed(lambda: R.batch_norm(x, bn, None, (e, bn2, key, lim)), "forward + bond statistics (keyed)")
    timed(lambda: R.batch_norm(x, bn, None, (e, bn2, None, None)), "forward + bond statistics (no key)")
    timed(lambda: R.update_running_stats(e, bn2, key, lim), "bond statistics alone")


def fb():
    x.grad = None
    out = R.batch_norm(x, bn)
    out.backward(g_out)


timed(fb, "forward + backward, no companion")
