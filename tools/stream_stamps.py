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
ap.add_argument("--blocks", type=int, default=0, help="grid cap of the forward (mkgnn_debug_set_grid_caps): 256 = one block per CU")
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
if args.blocks:
    _lib.check(lib.mkgnn_debug_set_grid_caps(args.blocks, 0, 0), "mkgnn_debug_set_grid_caps")
    for _ in range(3):
        Fn.kernelsetconv_details(x, plan, False, params, E, "auto")
lib.mkgnn_debug_set_stream_stamps.argtypes = [ctypes.c_void_p]      # (a bare int would be truncated to 32 bits)
buf = torch.zeros(512 * 4 * 16, dtype=torch.int64, device=dev)
lib.mkgnn_debug_set_stream_stamps(buf.data_ptr())
Fn.kernelsetconv_details(x, plan, False, params, E, "auto")
torch.cuda.synchronize()
lib.mkgnn_debug_set_stream_stamps(None)
s = buf.cpu().view(-1, 16)
s = s[s[:, 1] != 0]
# s_memtime counts per XCD from different origins: spans across waves are taken in s_memrealtime units (100 MHz, chip-wide),
# cycles only as differences inside one wave
life = (s[:, 1] - s[:, 0]).double()
rt0 = int(s[:, 12].min())
rt_span = int(s[:, 13].max()) - rt0
rt_life = (s[:, 13] - s[:, 12]).double()
clock = (life / rt_life.clamp(min=1)).mean().item() / 10.0      # cycles per 10 ns -> GHz
print(f"{s.shape[0]} waves; kernel span {rt_span / 100.0:.2f} us (s_memrealtime); in-kernel clock {clock:.3f} GHz (mean over waves of cycles / real time); "
      f"mean wave lifetime {life.mean().item():.0f} cycles = {rt_life.mean().item() / rt_span:.3f} of the span")
for g in sorted(set(s[:, 2].tolist())):
    m = s[s[:, 2] == g]
    start, end = (m[:, 12] - rt0).double() / 100.0, (m[:, 13] - rt0).double() / 100.0
    lf = (m[:, 1] - m[:, 0]).float()
    print(f"degree {g // 16} part {g % 16}: {m.shape[0]:4d} waves, {int(m[0, 3])} iterations; start {start.min():.2f}..{start.max():.2f} us; "
          f"end {end.min():.2f}..{end.max():.2f} us (mean {end.mean():.2f}); lifetime cycles mean {lf.mean():.0f} max {lf.max():.0f}")
    if int(m[:, 4:12].sum()) > 0:                    # make STAMPS=1: cycles per phase, mean over the group's waves
        ph = m[:, 4:12].float().mean(dim=0).tolist()
        print("        prologue %.0f  multiply %.0f  counted wait %.0f  barrier %.0f  DMA issue %.0f  epilogue: scan %.0f  bonds %.0f  mix+stores %.0f" %
              (ph[7], ph[0], ph[1], ph[2], ph[3], ph[5], ph[6], ph[4]))
