"""Where the host time of a shard-fed epoch goes (loader hand-off / static-buffer copy / graph replay), diagnostics."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import padding as P, shards as S                              # noqa: E402
from molkgnn_amd.receptive_field import attach_receptive_fields                # noqa: E402
from molkgnn_amd.synthetic import make_batch                                   # noqa: E402
from molkgnn_amd.train import GNNModel, backward, configure_optimizer          # noqa: E402

dev = torch.device("cuda:0")
B, nb = 4096, 16
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 3
raws = [make_batch(B, seed=50 + i, assay="all9", with_receptive_fields=False) for i in range(nb)]
model = GNNModel(num_layers=3).to(dev)
opt = configure_optimizer(model, lr=1e-3, capturable=True)
with tempfile.TemporaryDirectory() as d:
    paths = S.write_shards(d, raws)
    loader = S.ShardLoader(paths, B, device=dev, prefetch=3, workers=workers, fixed_shape=True, compact=True)
    csb = P.CompactStaticBatch(loader.shape, B, 28, 3, 7, dev)

    def step():
        csb.expand()
        attach_receptive_fields(csb.data, sizes=csb.data.bucket_sizes, overlap=True)
        model.zero_grad(set_to_none=True)
        loss = model.loss(csb.data)
        backward(loss)
        opt.step()

    csb.load(next(iter(loader)))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        model.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
    torch.cuda.current_stream().wait_stream(side)
    for cb in loader:
        csb.load(cb); g.replay()
    torch.cuda.synchronize()
    evs = [None, None]
    for mode in ("loader+load+replay", "loader+load+replay throttled", "loader+load", "loader only", "replay only"):
        t_next = t_load = t_rep = 0.0
        n = 0
        t0 = time.perf_counter()
        for _ in range(4):
            it = iter(loader) if mode != "replay only" else iter(range(nb))
            while True:
                t1 = time.perf_counter()
                cb = next(it, None)
                t2 = time.perf_counter()
                if cb is None:
                    break
                if mode.startswith("loader+load"):
                    csb.load(cb)
                t3 = time.perf_counter()
                if mode.endswith("throttled"):           # never more than two steps ahead of the GPU: the launch call does not block
                    if evs[n & 1] is not None:
                        evs[n & 1].synchronize()
                if mode.startswith("loader+load+replay") or mode == "replay only":
                    g.replay()
                if mode.endswith("throttled"):
                    evs[n & 1] = torch.cuda.Event()
                    evs[n & 1].record()
                t4 = time.perf_counter()
                t_next += t2 - t1; t_load += t3 - t2; t_rep += t4 - t3; n += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if mode.startswith("loader+load+replay") or mode == "replay only":      # GPU time of one replay while the loader runs
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gts = []
            for cb in (loader if mode != "replay only" else range(nb)):
                if mode != "replay only":
                    csb.load(cb)
                ea.record(); g.replay(); eb.record(); eb.synchronize()
                gts.append(ea.elapsed_time(eb))
            gts.sort()
            print(f"    one replay, timed with events while the loader runs: median {gts[len(gts) // 2]:.3f} ms")
        print(f"{mode:20s}: {1e3 * el / n:.3f} ms per batch  (host: next {1e3 * t_next / n:.3f}  load {1e3 * t_load / n:.3f}  replay {1e3 * t_rep / n:.3f})")
