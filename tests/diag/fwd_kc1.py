"""Diagnostic for the KC = 1 forward: which operand's upper k-lanes are wrong?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import kgnn_oracle as O
from molkgnn_amd import functional as Fn
from molkgnn_amd.kernels import KernelSetConv
from molkgnn_amd.plan import plan_from_data
from molkgnn_amd.synthetic import make_batch

dev = torch.device("cuda:0")
width, nm, counts = 16, 60, (0, 20, 0, 0)
cpu = make_batch(nm, seed=5)
bd = cpu.to(dev)
plan = plan_from_data(bd)
n = cpu.x.shape[0]
for mode in ("full", "x_hi_zero", "bank_hi_zero", "x_lo_zero"):
    torch.manual_seed(1)
    layer = KernelSetConv(*counts, D=3, node_attr_dim=width, edge_attr_dim=7)
    x_cpu = torch.randn(n, width, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        if mode == "x_hi_zero":
            x_cpu[:, 8:] = 0
        if mode == "x_lo_zero":
            x_cpu[:, :8] = 0
        if mode == "bank_hi_zero":
            for k in layer.trainable_kernelconv_set:
                if k is not None:
                    k.x_center[:, 8:] = 0
                    k.x_support[:, :, 8:] = 0
    state = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    store = torch.zeros(n, width, device=dev)
    store[:, :width] = x_cpu.to(dev)
    params, E = layer._bank_params("train", store)
    out, saved = Fn.kernelsetconv_details(store, plan, False, params, E, "mfma", raw=True)
    pr = saved[1][0].cpu()                      # [N_2, L, 4]: support, centre, edge, idx
    per = O.kernelset_params(state)
    sel, nei = cpu.selected_index_deg2, cpu.nei_index_deg2
    sc, table, idx, parts = O.kernelconv_cosmat(per[1], x_cpu[sel], cpu.p_focal_deg2, x_cpu[nei].reshape(-1, 2, width), cpu.nei_p_deg2,
                                                cpu.nei_edge_attr_deg2, False)
    dS = (pr[..., 0].T - parts["best"]).abs()
    dC = (pr[..., 1].T - parts["center"]).abs()
    dE = (pr[..., 2].T - parts["edge"]).abs()
    print(mode, "support max diff", float(dS.max()), "bad", int((dS > 1e-4).sum()), "/", dS.numel(),
          "| centre", float(dC.max()), int((dC > 1e-4).sum()), "| edge", float(dE.max()), int((dE > 1e-4).sum()),
          "| first bad atoms", (dC > 1e-4).any(0).nonzero().flatten()[:10].tolist(), flush=True)
