"""Diagnostic: cycle stamps inside the backward kernels (kc_backward_bank_lds, kc_backward_rows_mfma), per degree.
Needs a library built with the stamps compiled in: make -C molkgnn_amd/csrc clean && make -C molkgnn_amd/csrc STAMPS=1.
Per tile the kernel records: staging done, barrier passed, prefetch issued, accumulate loop done, barrier + id store."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MKGNN_SERIAL", "1")
from molkgnn_amd import _lib                         # noqa: E402
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
C.CDLL(_lib.LIB_PATH).mkgnn_debug_set_bwd_stamp_buffer.argtypes = [C.c_void_p]
setter = C.CDLL(_lib.LIB_PATH).mkgnn_debug_set_bwd_stamp_buffer
width = int(sys.argv[2]) if len(sys.argv) > 2 else 110
b = make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, seed=1798000).to(dev)
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7).to(dev)
store = torch.zeros(b.x.shape[0], width + (-width) % 4, device=dev)
store[:, :width] = torch.rand(b.x.shape[0], width, device=dev) * 2 - 1
x = store[:, :width].requires_grad_(True)
cot = torch.randn(b.x.shape[0], 110, device=dev)
for rep in range(4):
    if rep == 3:
        buf = torch.zeros(4 * 256 * 64, dtype=torch.int64, device=dev)
        assert setter(C.c_void_p(buf.data_ptr())) == 0
    out = layer._run(x, plan, False)
    (out * cot).sum().backward()
torch.cuda.synchronize()
setter(C.c_void_p(0))
st = buf.cpu().numpy().reshape(4, 256, 64)
for d in range(4):
    w = st[d]
    w = w[w[:, 0] != 0]
    if not len(w):
        continue
    t0 = w[:, 0].min()
    span = w[:, 62].max() - t0
    pro = np.median(w[:, 1] - w[:, 0])
    ntile = ((w[:, 2:62] != 0).sum(1) // 5)
    ph = [[] for _ in range(5)]
    for r in range(len(w)):
        prev = w[r, 1]
        for t in range(ntile[r]):
            for k in range(5):
                cur = w[r, 2 + 5 * t + k]
                ph[k].append(cur - prev)
                prev = cur
    med = [np.median(p) if p else 0 for p in ph]
    print(f"degree {d + 1}: blocks {len(w)} span {span} cyc; prologue {pro:.0f}; tiles/block {ntile.mean():.2f}; per tile med: "
          f"stage {med[0]:.0f} | barrier {med[1]:.0f} | prefetch issue {med[2]:.0f} | accumulate {med[3]:.0f} | barrier+ids {med[4]:.0f}"
          f" | sum {sum(med):.0f}; epilogue {np.median(w[:, 62] - w[np.arange(len(w)), 2 + 5 * ntile - 1]):.0f}")

# ---- rows kernel (kc_backward_rows_mfma): per wave {start, bank in LDS, then per tile: inputs ready, MFMAs done, stored}
rsetter = C.CDLL(_lib.LIB_PATH).mkgnn_debug_set_rows_stamp_buffer
rsetter.argtypes = [C.c_void_p]
rbuf = torch.zeros(4 * 1024 * 4 * 32, dtype=torch.int64, device=dev)
assert rsetter(C.c_void_p(rbuf.data_ptr())) == 0
out = layer._run(x, plan, False)
(out * cot).sum().backward()
torch.cuda.synchronize()
rsetter(C.c_void_p(0))
rs = rbuf.cpu().numpy().reshape(4, 1024 * 4, 32)
for d in range(4):
    w = rs[d]
    w = w[w[:, 0] != 0]
    if not len(w):
        continue
    t0 = w[:, 0].min()
    nt = ((w[:, 2:32] != 0).sum(1) // 3)
    last = np.array([w[r, 2 + 3 * nt[r] - 1] if nt[r] else w[r, 1] for r in range(len(w))])
    ph = [[] for _ in range(3)]
    for r in range(len(w)):
        prev = w[r, 1]
        for t in range(nt[r]):
            for k in range(3):
                cur = w[r, 2 + 3 * t + k]
                ph[k].append(cur - prev)
                prev = cur
    med = [np.median(p) if p else 0 for p in ph]
    print(f"rows degree {d + 1}: waves {len(w)} (with tiles: {(nt > 0).sum()}) span {last.max() - t0} cyc; start skew max {(w[:, 0] - t0).max()}; "
          f"bank copy med {np.median(w[:, 1] - w[:, 0]):.0f}; tiles/wave {nt.mean():.2f} max {nt.max()}; per tile med: inputs {med[0]:.0f} | "
          f"mfma {med[1]:.0f} | store {med[2]:.0f}")
