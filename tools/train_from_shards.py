"""The whole pipeline on one GPU, fed from packed shards: synthetic molecules with a label the graph determines (at least two
degree-4 atoms) -> shards on disk (molkgnn_amd/shards.py) -> ShardLoader(fixed_shape, compact): host-side padding, pinned
staging, one copy per batch -> CompactStaticBatch -> ONE captured graph per run (expand, receptive fields, index plan,
3-layer MolKGNN forward + backward with deferred bank gradients, FusedAdamW) replayed for every batch of every epoch.
The loss must fall and the held-out AUC must rise if every piece is right.
tools/train_from_shards.py [--molecules-per-shard 4096] [--shards 8] [--batch-size 1024] [--epochs 6]"""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import evaluation as E                                             # noqa: E402
from molkgnn_amd import padding as P                                                # noqa: E402
from molkgnn_amd import shards as S                                                 # noqa: E402
from molkgnn_amd.receptive_field import attach_receptive_fields                     # noqa: E402
from molkgnn_amd.synthetic import make_batch                                        # noqa: E402
from molkgnn_amd.train import GNNModel, backward, configure_optimizer               # noqa: E402


def labelled(n, seed):
    b = make_batch(n, seed=seed, assay="all9", with_receptive_fields=False)
    deg = torch.bincount(b.edge_index[0], minlength=b.x.shape[0])
    n4 = torch.zeros(n).index_add_(0, b.batch, (deg == 4).float())
    b.y = (n4 >= 2).float()
    return b


def run(molecules_per_shard=4096, n_shards=8, batch_size=1024, epochs=6, lr=3e-3, log=print):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as d:
        paths = S.write_shards(d, [labelled(molecules_per_shard, 100 + i) for i in range(n_shards)])
        loader = S.ShardLoader(paths, batch_size, device=dev, prefetch=3, workers=2, fixed_shape=True, compact=True)
        test = attach_receptive_fields(labelled(2048, 999).to(dev))
        model = GNNModel(num_layers=3).to(dev)
        opt = configure_optimizer(model, lr=lr, capturable=True)
        csb = P.CompactStaticBatch(loader.shape, batch_size, 28, 3, 7, dev)
        loss_box = []

        def step():
            csb.expand()
            attach_receptive_fields(csb.data, sizes=csb.data.bucket_sizes, overlap=True)
            model.zero_grad(set_to_none=True)
            loss = model.loss(csb.data)
            backward(loss)
            opt.step()
            return loss

        def evaluate():
            model.eval()
            with torch.no_grad():
                pred, _ = model(test)
            model.train()
            return float(E.calculate_logAUC(test.y, pred.view(-1))), float(E.calculate_auc(test.y, pred.view(-1)))

        before = evaluate()
        csb.load(next(iter(loader)))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()                                   # (two real steps on the first batch: warm-up before the capture)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                loss_box.append(step())
        torch.cuda.current_stream().wait_stream(side)
        losses = []
        t0 = time.perf_counter()
        n = 0
        for ep in range(epochs):
            for cb in loader:
                csb.load(cb)
                g.replay()
                n += 1
            losses.append(float(loss_box[0].detach()))   # (the last batch's loss: one host read per epoch)
            log(f"epoch {ep}: loss of its last batch {losses[-1]:.4f}")
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        after = evaluate()
        log(f"{n} steps of {batch_size} molecules in {el:.2f} s ({n * batch_size / el / 1e6:.2f} M molecules/s from shards); "
            f"held-out logAUC {before[0]:.3f} -> {after[0]:.3f}, AUC {before[1]:.3f} -> {after[1]:.3f}")
        loader.close()
        return losses, before, after


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules-per-shard", type=int, default=4096)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=6)
    a = ap.parse_args()
    run(a.molecules_per_shard, a.shards, a.batch_size, a.epochs)
