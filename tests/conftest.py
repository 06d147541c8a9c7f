import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _poison_free_gpu_memory(request):
    """Before every GPU test, fill the caching allocator's free blocks with NaN bit patterns: a kernel that reads a
    ``torch.empty`` buffer it never wrote (workspace, padding, an absent degree's slab ...) then produces NaNs or
    wild indices instead of passing by luck on zero-filled fresh memory."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(4)]      # 4 x 256 MB
        small = [torch.full((n,), float("nan"), device="cuda") for n in (128, 1024, 8192, 65536, 1 << 20) for _ in range(8)]
        torch.cuda.synchronize()
        del junk, small
    yield


@pytest.fixture(autouse=True)
def _per_operator_path_unless_asked(request, monkeypatch):
    """The molecule-resident small-batch kernels (molkgnn_amd.molecule) take small batches by default; the tests of the
    per-operator kernels pin that path (``tests/test_molecule.py`` switches the molecule-resident one on itself)."""
    if request.node.get_closest_marker("gpu") is None or "test_molecule" in request.node.nodeid:
        yield
        return
    from molkgnn_amd import molecule
    monkeypatch.setattr(molecule, "_MODE", "0")
    yield
