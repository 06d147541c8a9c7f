"""End-to-end sanity run of the whole path on one GPU: synthetic molecules -> HIP receptive-field builder -> 3-layer
MolKGNN (HIP convolution, propagate, batch norm, readout, head + loss) -> AdamW, then the reference's metrics.
The label is a structural property the network can read off the graph (at least two degree-4 atoms), so the loss
must fall and logAUC must rise if forward and backward are right."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import evaluation as E                                             # noqa: E402
from molkgnn_amd.receptive_field import attach_receptive_fields                    # noqa: E402
from molkgnn_amd.synthetic import make_batch                                       # noqa: E402
from molkgnn_amd.train import GNNModel, configure_optimizer                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--molecules", type=int, default=2048)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--lr", type=float, default=3e-3)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def labelled(seed):
    b = make_batch(args.molecules, seed=seed, with_receptive_fields=False).to(dev)
    attach_receptive_fields(b)                                                      # HIP builder (GPU batch)
    deg = torch.bincount(b.edge_index[0], minlength=b.x.shape[0])
    n4 = torch.zeros(args.molecules, device=dev).index_add_(0, b.batch, (deg == 4).float())
    b.y = (n4 >= 2).long()
    return b


train, test = [labelled(100 + i) for i in range(4)], labelled(999)
model = GNNModel(num_layers=3).to(dev)
opt = configure_optimizer(model, lr=args.lr, fused=True)


def evaluate(b):
    model.eval()
    with torch.no_grad():
        pred, _ = model(b)
    model.train()
    return E.calculate_logAUC(b.y, pred.view(-1)), E.calculate_auc(b.y, pred.view(-1))


print(f"positives: {float(test.y.float().mean()):.3f}; before: logAUC {evaluate(test)[0]:.3f} AUC {evaluate(test)[1]:.3f}")
t0 = time.perf_counter()
for step in range(args.steps):
    b = train[step % len(train)]
    opt.zero_grad(set_to_none=True)
    loss = model.loss(b)
    loss.backward()
    opt.step()
    if step % 25 == 0 or step == args.steps - 1:
        print(f"step {step:4d} loss {float(loss.detach()):.4f}")
torch.cuda.synchronize()
la, au = evaluate(test)
print(f"after {args.steps} steps ({time.perf_counter() - t0:.1f} s): held-out logAUC {la:.3f} AUC {au:.3f}")
