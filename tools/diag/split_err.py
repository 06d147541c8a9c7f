"""Where the split-fp16 forward differs from float64 (diagnostics of tests/test_scale_parity.py::test_split_fp16_products_are_fp32_grade)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd import _lib                        # noqa: E402
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 110
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
spread = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(3)
b = make_batch(384, seed=1798321).to(dev)
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
store = torch.zeros(b.x.shape[0], width + (-width) % 4, device=dev)
store[:, :width] = (torch.rand(b.x.shape[0], width, device=dev) * 2 - 1) * scale
if spread:
    store[:, :width] *= torch.exp2(torch.randint(-6, 1, (b.x.shape[0], width), device=dev).float())
    store[::97] *= 1e-3
x = store[:, :width]


def unit64(t):
    t = t.double()
    return t / t.norm(dim=-1, keepdim=True).clamp_min(1e-8)


xu = unit64(x)
for mode in (0, 1):
    lib.mkgnn_debug_set_forward_products(mode)
    out, saved = Fn.kernelsetconv_details(x, plan, False, params, E, "auto")
    torch.cuda.synchronize()
    for d in range(4):
        bk = plan.buckets[d]
        if not bk.count:
            continue
        cen64 = unit64(params[7 * d].detach()) @ xu[bk.sel].t()
        sc = saved[d][1].double()
        e = (sc[1] - cen64).abs()
        l, n = divmod(int(e.argmax()), e.shape[1])
        bad = (e.max(dim=0).values > 1e-6).nonzero().reshape(-1)
        print(f"mode {mode} degree {d + 1}: centre max err {float(e.max()):.3g} at kernel {l} atom position {n} of {bk.count} "
              f"(row norm {float(x[bk.sel[n]].norm()):.3g}); positions off by > 1e-6: {bad.numel()} first {bad[:10].tolist()} "
              f"row norms {[round(float(x[bk.sel[i]].norm()), 5) for i in bad[:6].tolist()]}")
        if d == 0 and mode == 1:
            r = (sc[1] / cen64)
            print("        centre got/expected, kernels 0..3 x atoms 0..5 and 700..705:", [[round(float(r[l, n]), 5) for n in (0, 1, 2, 3, 4, 5, 700, 701, 702, 703)] for l in range(3)])
        if d == 0:
            sup64 = unit64(params[1].detach()[:, 0]) @ xu[bk.nei.reshape(bk.count, -1)[:, 0]].t()
            e = (sc[0] - sup64).abs()
            bad = (e.max(dim=0).values > 1e-6).nonzero().reshape(-1)
            print(f"        support max err {float(e.max()):.3g}; positions off: {bad.numel()} first {bad[:10].tolist()}")
lib.mkgnn_debug_set_forward_products(-1)
