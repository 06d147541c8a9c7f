"""Per-group wave lifetimes of kc_forward_stream (diagnostics): start / end cycle stamps of every wave."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib                        # noqa: E402
from molkgnn_amd import functional as Fn            # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data         # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch-size", type=int, default=4096)
ap.add_argument("--width", type=int, default=110)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
b = make_batch(args.batch_size, seed=1798000).to(dev)
plan = plan_from_data(b)
layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=args.width, edge_attr_dim=7).to(dev)
params, E = layer._bank_params("train", b.x)
F = args.width
store = torch.zeros(b.x.shape[0], F + (-F) % 4, device=dev)
store[:, :F] = torch.rand(b.x.shape[0], F, device=dev) * 2 - 1
x = store[:, :F]
for _ in range(5):
    Fn.kernelsetconv_details(x, plan, False, params, E, "auto")
import ctypes
lib = _lib.load()
lib.mkgnn_debug_set_stream_stamps.argtypes = [ctypes.c_void_p]      # (a bare int would be truncated to 32 bits)
buf = torch.zeros(512 * 4 * 16, dtype=torch.int64, device=dev)
lib.mkgnn_debug_set_stream_stamps(buf.data_ptr())
Fn.kernelsetconv_details(x, plan, False, params, E, "auto")
torch.cuda.synchronize()
lib.mkgnn_debug_set_stream_stamps(None)
s = buf.cpu().view(-1, 16)
s = s[s[:, 1] != 0]
t0 = int(s[:, 0].min())
print(f"{s.shape[0]} waves; kernel span {int(s[:, 1].max()) - t0} ticks (s_memtime / readcyclecounter units)")
for g in sorted(set(s[:, 2].tolist())):
    m = s[s[:, 2] == g]
    start, end = m[:, 0] - t0, m[:, 1] - t0
    life = (end - start).float()
    print(f"degree {g // 16} part {g % 16}: {m.shape[0]:4d} waves, {int(m[0, 3])} iterations; start {int(start.min())}..{int(start.max())}; "
          f"lifetime mean {life.mean():.0f} max {life.max():.0f}")
    if int(m[:, 4:12].sum()) > 0:                    # make STAMPS=1: cycles per phase, mean over the group's waves
        ph = m[:, 4:12].float().mean(dim=0).tolist()
        print("        multiply %.0f  counted wait %.0f  barrier %.0f  DMA issue %.0f  epilogue: scan %.0f  bonds %.0f  mix+stores %.0f" %
              (ph[0], ph[1], ph[2], ph[3], ph[5], ph[6], ph[4]))
