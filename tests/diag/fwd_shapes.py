"""Diagnostic: per-degree forward mismatches of the streamed kernel against the oracle for (counts, width, molecules)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import kgnn_oracle as O
from molkgnn_amd import functional as Fn
from molkgnn_amd.kernels import KernelSetConv
from molkgnn_amd.plan import plan_from_data
from molkgnn_amd.synthetic import make_batch

dev = torch.device("cuda:0")
cases = [((10, 20, 30, 50), 16, 100), ((10, 10, 10, 10), 16, 100), ((10, 20, 0, 0), 16, 100), ((0, 0, 30, 0), 16, 100), ((0, 20, 0, 0), 16, 100)]
for counts, width, nm in cases:
    cpu = make_batch(nm, seed=77 + width)
    bd = cpu.to(dev)
    plan = plan_from_data(bd)
    torch.manual_seed(width)
    layer = KernelSetConv(*counts, D=3, node_attr_dim=width, edge_attr_dim=7)
    state = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    n = cpu.x.shape[0]
    x_cpu = torch.randn(n, width, generator=torch.Generator().manual_seed(width))
    store = torch.zeros(n, width + (-width) % 4, device=dev)
    store[:, :width] = x_cpu.to(dev)
    params, E = layer._bank_params("train", store[:, :width])
    out, saved = Fn.kernelsetconv_details(store[:, :width], plan, False, params, E, "mfma")
    out = out.cpu()
    ref = O.kernelsetconv(O.kernelset_params(state), x_cpu, cpu, False, form="cosmat")
    off = 0
    msg = []
    for d in range(1, 5):
        sel = getattr(cpu, f"selected_index_deg{d}")
        L = counts[d - 1]
        diff = (out[sel][:, off:off + L] - ref[sel][:, off:off + L]).abs()
        badrows = (diff > 1e-3).any(dim=1)
        badcols = (diff > 1e-3).any(dim=0).nonzero().flatten().tolist()
        msg.append(f"d{d}: {int((diff > 1e-3).sum())}/{diff.numel()} bad, rows {int(badrows.sum())}/{sel.numel()}, cols {badcols[:6]}..{badcols[-3:]}")
        off += L
    print(counts, width, nm, " | ".join(msg), flush=True)
