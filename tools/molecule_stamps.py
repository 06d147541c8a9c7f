"""Phase cycle stamps of chunk 0 of the molecule-resident step kernel (mkgnn_debug_molecule_stamps):
python3 tools/molecule_stamps.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib                                                   # noqa: E402
from molkgnn_amd.synthetic import make_batch                                   # noqa: E402
from molkgnn_amd.train import GNNModel                                         # noqa: E402
from molkgnn_amd.train import backward as train_backward                       # noqa: E402

from molkgnn_amd import molecule as _mol                                        # noqa: E402
_mol._MODE = "1"                                                               # (the molecule-resident path is opt-in)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
torch.manual_seed(1798)
model = GNNModel(ffn_dropout_rate=0.25).to(dev).train()
b = make_batch(B, seed=435008000 + 700, assay="435008").to(dev)
lib = _lib.load()
import ctypes                                                                  # noqa: E402
lib.mkgnn_debug_molecule_stamps.argtypes = [ctypes.c_void_p]                   # (an untyped int would be truncated to 32 bits)
lib.mkgnn_debug_molecule_stamps.restype = ctypes.c_int
buf = torch.zeros(256, dtype=torch.int64, device=dev)
for it in range(3):
    model.zero_grad(set_to_none=True)
    if it == 2:
        lib.mkgnn_debug_molecule_stamps(buf.data_ptr())
    train_backward(model.loss(b))
torch.cuda.synchronize()
lib.mkgnn_debug_molecule_stamps(None)
st = buf.cpu().tolist()
n = max(i for i, v in enumerate(st) if v) + 1
names = ["start", "meta", "bn"]
for li in range(3):
    names += [f"f{li} norms", f"f{li} gemm0", f"f{li} pairs0", f"f{li} gemm1", f"f{li} pairs1", f"f{li} propagate"]
names += ["readout fwd", "head", "readout bwd"]
for li in (2, 1, 0):
    names += [f"b{li} pairs", f"b{li} rows+edge", f"b{li} Cf0", f"b{li} gemm0", f"b{li} Cf1", f"b{li} gemm1", f"b{li} project"]
names += ["end"]
print(f"batch {B}: {n} stamps, total {st[n - 1] - st[0]} cycles")
for i in range(1, n):
    print(f"{names[i] if i < len(names) else i:18s} {st[i] - st[i - 1]:8d}")
