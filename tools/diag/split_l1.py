import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd.synthetic import make_batch
from molkgnn_amd.train import GNNModel
from molkgnn_amd import functional as Fn, KernelLayer as KL
from molkgnn_amd.plan import plan_from_data
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = GNNModel().to(dev); m.train()
b = make_batch(512, seed=3).to(dev)
plan = plan_from_data(b)
l0 = m.gnn_model.gnn.layers[0] if hasattr(m, "gnn_model") else None
print("layers0", type(l0).__name__, "ROWS_SPLIT", KL._ROWS_SPLIT)
print("can_prepare", l0._can_prepare(), "variant", l0.variant, l0.backward_variant)
params, E = l0._bank_params("train", b.x)
print("F", params[0].shape, "E", E, [int(p.shape[0]) for p in params[0::7]])
print("supported", Fn.rows_split_supported(plan, params, int(params[0].shape[1]), E, plan.n_atoms))
l1 = m.gnn_model.gnn.layers[1]
params1, E1 = l1._bank_params("train", b.x)
print("layer1 supported", Fn.rows_split_supported(plan, params1, int(params1[0].shape[1]), E1, plan.n_atoms))
import molkgnn_amd._lib as L
print("last error:", L.load().mkgnn_last_error() if hasattr(L.load(), "mkgnn_last_error") else None)
import molkgnn_amd.readout as R
orig_bn = R.batch_norm
def spy_bn(*a, **k):
    print("batch_norm split_out =", k.get("split_out"))
    out = orig_bn(*a, **k)
    print("  tagged:", Fn.is_rows_split(out), out.shape, out.stride())
    return out
R.batch_norm = spy_bn
orig_fi = Fn._forward_impl
def spy_fi(*a, **k):
    r = orig_fi(*a, **k)
    print("  _forward_impl x_split =", r[-1] if isinstance(r, tuple) else r, "x tagged", Fn.is_rows_split(a[0]) if torch.is_tensor(a[0]) else None)
    return r
Fn._forward_impl = spy_fi
for bs in (512, 4096):
    bb = make_batch(bs, seed=5).to(dev)
    print("batch", bs)
    loss = m.loss(bb)
    loss.backward()
torch.cuda.synchronize()
