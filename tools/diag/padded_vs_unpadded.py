"""Diagnostic: where do a padded and an unpadded run of the same molecules first differ?  (stage by stage)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from molkgnn_amd import padding as P
from molkgnn_amd import readout as R
from molkgnn_amd.plan import plan_from_data
from molkgnn_amd.receptive_field import attach_receptive_fields
from molkgnn_amd.synthetic import make_batch
from molkgnn_amd.train import GNNModel

dev = torch.device("cuda:0")
torch.manual_seed(17)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
raws = [make_batch(B, seed=4000 + i, with_receptive_fields=False) for i in range(2)]
shape = P.fixed_shape([P.degree_histogram(r) for r in raws])
model = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev)
net = model.gnn_model
r = raws[0]
ub = attach_receptive_fields(r.to(dev))
pb = P.pad_batch(r, shape, B).to(dev)
attach_receptive_fields(pb, sizes=pb.bucket_sizes)
n = r.x.shape[0]
print("atoms", n, "padded", pb.x.shape[0], "shape", shape)
stages = []
for b in (ub, pb):
    out = {}
    with torch.no_grad():
        x = R.batch_norm(b.x, net.node_batch_norm, getattr(b, 'n_valid_atoms', None))
        out["bn"] = x.clone()
        plan = plan_from_data(b)
        h = x
        for i, layer in enumerate(net.gnn.layers):
            sim = layer._run(h, plan, i == 2, False, block_rows=False, fuse_propagate=False)
            out[f"sim{i}"] = sim.clone()
            from molkgnn_amd import functional as Fn
            h = Fn.propagate_add(sim, plan, out_pad=(-sim.shape[1]) % 4)
            out[f"h{i}"] = h.clone()
    stages.append(out)
    for d in range(1, 5):
        print("bucket", d, getattr(b, f"selected_index_deg{d}").shape[0], end="; ")
    print()
for k in stages[0]:
    a, c = stages[0][k][:n].float(), stages[1][k][:n].float()
    d = (a - c).abs()
    rows = (d > 0).any(dim=1)
    print(f"{k:6s} shape {tuple(a.shape)} max|diff| {float(d.max()):.3e} elements differing {int((d > 0).sum())} rows differing {int(rows.sum())}"
          f" > 1e-3: {int((d > 1e-3).sum())}  first rows {rows.nonzero().flatten()[:8].tolist()} last {rows.nonzero().flatten()[-4:].tolist()}")
# which degree do the differing rows of sim0 have?
d0 = (stages[0]["sim0"][:n] - stages[1]["sim0"][:n]).abs()
rows = (d0 > 0).any(dim=1)
for d in range(1, 5):
    sel = getattr(ub, f"selected_index_deg{d}")
    selp = getattr(pb, f"selected_index_deg{d}")
    same = torch.equal(sel, selp[:sel.shape[0]])
    print("degree", d, "atoms", sel.shape[0], "differing", int(rows[sel].sum()), "bucket prefix equal", same,
          "nei equal", torch.equal(getattr(ub, f"nei_index_deg{d}"), getattr(pb, f"nei_index_deg{d}")[:sel.shape[0] * d]))
    if rows[sel].any():
        pos = rows[sel].nonzero().flatten()
        print("   positions in bucket: first", pos[:6].tolist(), "last", pos[-6:].tolist(), "count", pos.numel())
