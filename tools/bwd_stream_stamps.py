"""Per-phase cycle stamps of the streamed backward kernels (kc_backward_bank_stream, kc_backward_rows_stream), per degree
group: needs a build with the stamps compiled in --
    make -C molkgnn_amd/csrc VARIANT=stamps STAMPS=1 -j6
    MKGNN_LIB=build_variants/stamps/libmolkgnn_hip.so python3 tools/bwd_stream_stamps.py [batch] [width]
Mean cycles per wave and phase (the stamps themselves cost about a tenth)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MKGNN_SERIAL", "1")                 # one kernel at a time: a wave's phases are its own
from molkgnn_amd import _lib                         # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402
from molkgnn_amd import functional as Fn             # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
width = int(sys.argv[2]) if len(sys.argv) > 2 else 110
dev = torch.device("cuda:0")
lib = _lib.load()
for nm in ("mkgnn_debug_set_bank_stream_stamps", "mkgnn_debug_set_rows_stream_stamps"):
    getattr(lib, nm).argtypes = [ctypes.c_void_p]
    getattr(lib, nm).restype = ctypes.c_int
torch.manual_seed(0)
b = make_batch(B, seed=1798000).to(dev)
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
store = torch.zeros(b.x.shape[0], width + (-width) % 4, device=dev)
store[:, :width] = torch.rand(b.x.shape[0], width, device=dev) * 2 - 1
wgt = torch.randn(b.x.shape[0], 110, device=dev)


def step():
    x = store[:, :width].detach().requires_grad_(True)
    for p in params:
        p.grad = None
    h = Fn.kernelsetconv(x, plan, False, params, E, "auto", block_rows=True, propagate=True, backward_variant="fast")
    (h * wgt).sum().backward()
    torch.cuda.synchronize()


for _ in range(3):
    step()
names = {"bank": ("tile coefficients", "operand preparation", "row products", "counted wait", "barrier", "DMA issue", "slab store"),
         "rows": ("coefficients + next loads", "matrix products", "partial tile to LDS", "barrier", "finishing pass", "prologue (before the lifetime)")}
for which in ("bank", "rows"):
    buf = torch.zeros(1024 * 4 * 16, dtype=torch.int64, device=dev)
    setter = getattr(lib, f"mkgnn_debug_set_{which}_stream_stamps")
    if setter(buf.data_ptr()) != 0:
        raise SystemExit("this library was built without the stamps: make VARIANT=stamps STAMPS=1 and MKGNN_LIB=...")
    step()
    setter(None)
    s = buf.cpu().view(-1, 16)
    s = s[(s[:, 1] != 0) & (s[:, 0] != 0)]
    t0 = int(s[:, 0].min())
    print(f"kc_backward_{which}_stream<{(width + 15) // 16}>: {s.shape[0]} waves, span {int(s[:, 1].max()) - t0} cycles, batch {B}, F = {width}")
    print("| degree (part) | waves | tiles per stream | " + " | ".join(names[which]) + " | lifetime |")
    print("|---|---|---|" + "---|" * (len(names[which]) + 1))
    for g in sorted(set(s[:, 2].tolist())):
        m = s[s[:, 2] == g]
        life = (m[:, 1] - m[:, 0]).float().mean()
        ph = m[:, 4:4 + len(names[which])].float().mean(dim=0).tolist()
        print(f"| {g // 16} ({g % 16}) | {m.shape[0]} | {int(m[0, 3])} | " + " | ".join(f"{v / 1e3:.1f} k" for v in ph) + f" | {life / 1e3:.1f} k |")
