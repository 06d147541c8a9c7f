"""End-to-end sanity of the reference's own regime (README.md:81: batch 16): the same seeded training run -- 3-layer MolKGNN,
AdamW, batches of 16 synthetic molecules, the structural label of tools/train_synthetic.py -- through the molecule-resident
one-launch step (default up to 32 molecules) and through the per-operator kernels (MKGNN_MOLECULE=0), then the reference's
metrics on a held-out batch of 2048.  The two loss curves must track each other and both must learn.
    python3 tools/train_small_batches.py [--steps 600]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import evaluation as E                                             # noqa: E402
from molkgnn_amd import molecule as M                                               # noqa: E402
from molkgnn_amd.synthetic import make_batch                                       # noqa: E402
from molkgnn_amd.train import GNNModel, configure_optimizer                        # noqa: E402
from molkgnn_amd.train import backward as train_backward                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--lr", type=float, default=3e-3)
args = ap.parse_args()
dev = torch.device("cuda:0")


def labelled(n, seed):
    b = make_batch(n, seed=seed).to(dev)
    b.num_graphs = n
    deg = torch.bincount(b.edge_index[0], minlength=b.x.shape[0])
    n4 = torch.zeros(n, device=dev).index_add_(0, b.batch, (deg == 4).float())
    b.y = (n4 >= 2).float()
    return b


train = [labelled(args.batch, 5000 + i) for i in range(256)]
test = labelled(2048, 999)
curves = {}
for mode in ("", "0"):
    M._MODE = mode
    torch.manual_seed(0)
    model = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev).train()
    opt = configure_optimizer(model, lr=args.lr, fused=True)
    losses = []
    t0 = time.perf_counter()
    for step in range(args.steps):
        b = train[step % len(train)]
        opt.zero_grad(set_to_none=True)
        loss = model.loss(b)
        train_backward(loss)
        opt.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    with torch.no_grad():
        pred, _ = model(test)
    la, au = E.calculate_logAUC(test.y.long(), pred.view(-1)), E.calculate_auc(test.y.long(), pred.view(-1))
    ls = torch.stack(losses).cpu()
    curves[mode] = ls
    name = "molecule-resident step" if mode == "" else "per-operator kernels  "
    print(f"{name}: loss {float(ls[:20].mean()):.4f} -> {float(ls[-50:].mean()):.4f} over {args.steps} eager steps of batch {args.batch} "
          f"({1e3 * dt / args.steps:.2f} ms per eager step), held-out AUC {au:.3f} logAUC {la:.3f}")
d = (curves[""] - curves["0"]).abs()
print(f"loss curves: max |difference| over the first 50 steps {float(d[:50].max()):.2e}, over all steps {float(d.max()):.2e} "
      "(tied neighbour orders are resolved by rounding, differently in the two paths: the runs drift apart slowly)")
