"""Forward kernel alone: warm (back-to-back) vs cold (1 GiB written between launches), widths 28 and 110, pre-split rows."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molkgnn_amd import _lib, functional as Fn
from molkgnn_amd.kernels import KernelSetConv
from molkgnn_amd.plan import plan_from_data
from molkgnn_amd.synthetic import make_batch
dev = torch.device("cuda:0")
lib = _lib.load()
lib.mkgnn_debug_last_fused_forward_ms.restype = ctypes.c_float
b = make_batch(4096, seed=1798000).to(dev)
plan = plan_from_data(b)
junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)     # 1 GiB
for width in (28, 110):
    torch.manual_seed(0)
    layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7).to(dev)
    params, E = layer._bank_params("train", b.x)
    store = torch.zeros(b.x.shape[0], width + (-width) % 4, device=dev)
    store[:, :width] = torch.rand(b.x.shape[0], width, device=dev) * 2 - 1
    xs = Fn.presplit_rows(store[:, :width])
    x = xs.detach()
    setattr(x, Fn._INV_ATTR, (getattr(xs, Fn._INV_ATTR)[0], x._version))
    Fn.mark_rows_split(x)
    for _ in range(3):
        Fn.kernelsetconv_details(x, plan, False, params, E, "auto", raw=True)
    res = {}
    for mode in ("warm", "cold"):
        lib.mkgnn_debug_time_fused_forward(1)
        s = []
        for _ in range(12):
            if mode == "cold":
                junk.fill_(1.0)
            Fn.kernelsetconv_details(x, plan, False, params, E, "auto", raw=True)
            s.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
        lib.mkgnn_debug_time_fused_forward(0)
        res[mode] = sorted(s)[len(s) // 2]
    print(f"width {width}: single-bracket launch warm {1e3 * res['warm']:.1f} us, cold {1e3 * res['cold']:.1f} us")
