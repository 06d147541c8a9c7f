"""Wall-clock breakdown of one training step (HIP events): forward / backward / optimiser, and per-op pieces."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import functional as Fn
from molkgnn_amd.plan import plan_from_data
from molkgnn_amd.synthetic import make_batch
from molkgnn_amd.train import GNNModel, configure_optimizer

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
model = GNNModel().to(dev).train()
opt = configure_optimizer(model)
b = make_batch(B, seed=1).to(dev)
plan = plan_from_data(b); _ = plan.scatter, plan.csr_in, plan.csr_out

def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def fwd():
    return model.loss(b)
def fwd_bwd():
    model.zero_grad(set_to_none=True); model.loss(b).backward()
def full():
    fwd_bwd(); opt.step()
with torch.no_grad():
    t_inf = timed(lambda: model.loss(b))
t_f = timed(fwd); t_fb = timed(fwd_bwd); t_all = timed(full)
print(f"batch {B}: inference fwd {t_inf:.3f} ms | training fwd {t_f:.3f} | fwd+bwd {t_fb:.3f} | +AdamW {t_all:.3f} ms")
# pieces
gnn = model.gnn_model
x0 = gnn.node_batch_norm(b.x).detach()
layer1 = gnn.gnn.layers[1]
params, E = layer1._bank_params("train", x0)
store = torch.zeros(b.x.shape[0], 112, device=dev); store[:, :110] = torch.rand(b.x.shape[0], 110, device=dev)
h = store[:, :110].requires_grad_(True)
def lay_f():
    return Fn.kernelsetconv(h, plan, False, params, E)
out = lay_f(); g = torch.rand_like(out)
def lay_fb():
    o = Fn.kernelsetconv(h, plan, False, params, E); torch.autograd.grad(o, [h] + [p for p in params if p.requires_grad], g, allow_unused=True)
print(f"N-hop layer: fwd(train) {timed(lay_f):.3f} ms | fwd+bwd {timed(lay_fb):.3f} ms")
sim = out.detach().requires_grad_(True)
def prop_fb():
    o = Fn.propagate_add(sim, plan, out_pad=2); torch.autograd.grad(o, sim, torch.ones_like(o))
print(f"propagate fwd+bwd {timed(prop_fb):.3f} ms")
hh = torch.rand(b.x.shape[0], 110, device=dev, requires_grad=True)
def readout_fb():
    z = gnn.graph_embedding_lin2(gnn.dropout(gnn.act(gnn.graph_embedding_lin1(hh))))
    p = gnn.pool(z, b.batch, b.num_graphs); p.sum().backward()
print(f"readout (lin1/swish/lin2/pool) fwd+bwd {timed(readout_fb):.3f} ms")
xx = b.x.clone().requires_grad_(True)
def bn_fb():
    gnn.node_batch_norm(xx).sum().backward()
print(f"batchnorm fwd+bwd {timed(bn_fb):.3f} ms; optimizer step {timed(lambda: opt.step()):.3f} ms")
