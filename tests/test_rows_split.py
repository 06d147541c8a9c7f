"""Pre-split rows (round 6; csrc/kgnn_split.h, functional.ROWS_SPLIT): a tensor that only the next kernel convolution reads
-- h = propagate(sim_sc) between two layers, reference KernelLayer.py:119-123 -> kernels.py:527,543 -- is written by its
producer as the fp16 hi | lo halves the streamed kernels' matrix instructions take.  ``pytest -m gpu``.

What must hold, and is checked here against the SAME operators on ordinary fp32 rows (which the rest of the suite pins to the
oracle and to the reference's golden numbers):

* the producer's rows decode to the fp32 rows within 2^-22 of an element (hi + lo is exact, the split drops at most two bits);
* the forward on pre-split rows is BIT FOR BIT the forward on fp32 rows -- scores, pair records, chirality signs (the same
  split of the same scaled value, made once instead of in every wave);
* the bank gradients are bit for bit the same (the bank kernel's B operand is the same pair of halves); grad x differs only
  through the row the gather un-normalises with (x to 2^-22);
* a 3-layer network gives the same embedding and the same gradients whether h travels pre-split or not, eager and inside
  a captured graph, in training and in inference.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    return torch.device("cuda:0")


def _decode(h_split: torch.Tensor, inv: torch.Tensor, width: int) -> torch.Tensor:
    """fp32 values of pre-split rows: (hi + lo) / 2^(exponent(inv) + 8)."""
    n = h_split.shape[0]
    store = torch.as_strided(h_split, (n, h_split.stride(0)), (h_split.stride(0), 1))
    w4 = (width + 3) // 4 * 4
    halves = store[:, :w4].contiguous().view(torch.float16).view(n, w4 // 4, 2, 4).float()      # [n, granule, hi|lo, 4]
    vals = (halves[:, :, 0, :] + halves[:, :, 1, :]).reshape(n, w4)[:, :width]
    e = (inv.view(torch.int32) >> 23) & 0xFF
    scale = torch.exp2((e - 127 + 8).float())
    return vals / scale[:, None]


def _setup(width, n_mol=700, seed=31, dup=0.1):
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(11)
    b = make_batch(n_mol, seed=seed, duplicate_fraction=dup).to(dev)
    plan = plan_from_data(b)
    first = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7).to(dev)
    second = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=110, edge_attr_dim=7).to(dev)
    n = b.x.shape[0]
    store = torch.zeros(n, width + (-width) % 4, device=dev)
    store[:, :width] = torch.randn(n, width, device=dev)
    return dev, b, plan, first, second, store


@pytest.mark.parametrize("width", [28, 110])
def test_producer_writes_rows_that_decode_to_the_fp32_rows(width):
    from molkgnn_amd import functional as Fn
    dev, b, plan, first, _, store = _setup(width)
    params, E = first._bank_params("train", b.x)
    x = store[:, :width]
    with torch.no_grad():
        h = Fn.kernelsetconv(x, plan, False, params, E, "auto", block_rows=True, propagate=True)
        hs = Fn.kernelsetconv(x, plan, False, params, E, "auto", block_rows=True, propagate="split")
    assert Fn.is_rows_split(hs) and not Fn.is_rows_split(h)
    inv, inv_s = getattr(h, Fn._INV_ATTR)[0], getattr(hs, Fn._INV_ATTR)[0]
    assert torch.equal(inv, inv_s)
    back = _decode(hs, inv_s, 110)
    err = (back - h).abs()
    # two roundings to nearest of the SCALED value: 2^-22 of the element, or -- an element below 2^-3 scaled, i.e. 2^-11 of its
    # row's norm, whose lo half is an fp16 subnormal -- 2^-25 scaled = 2^-33 of the row's norm
    e = (inv_s.view(torch.int32) >> 23) & 0xFF
    scale = torch.exp2((e - 127 + 8).float())[:, None]
    assert bool((err * scale <= 2.0 ** -22 * h.abs() * scale + 2.0 ** -25).all()), float((err / h.abs().clamp_min(1e-30)).max())
    assert float((err / h.abs().clamp_min(1e-30))[h.abs() * scale >= 0.125].max()) <= 2.0 ** -22
    # an in-place change of the tensor drops the tag (its values are no longer what the producer wrote)
    hs.add_(0.0)
    assert not Fn.is_rows_split(hs)


@pytest.mark.parametrize("last", [False, True])
def test_forward_on_presplit_rows_is_bit_for_bit_the_forward_on_fp32_rows(last):
    from molkgnn_amd import functional as Fn
    dev, b, plan, first, second, store = _setup(28, dup=0.2)
    p1, E = first._bank_params("train", b.x)
    p2, _ = second._bank_params("train", b.x)
    assert Fn.rows_split_supported(plan, p2, 110, E, plan.n_atoms)
    with torch.no_grad():
        h = Fn.kernelsetconv(store[:, :28], plan, False, p1, E, "auto", block_rows=True, propagate=True)
        hs = Fn.kernelsetconv(store[:, :28], plan, False, p1, E, "auto", block_rows=True, propagate="split")
        out, saved = Fn.kernelsetconv_details(h, plan, last, p2, E, raw=True)
        out_s, saved_s = Fn.kernelsetconv_details(hs, plan, last, p2, E, raw=True)
    assert torch.equal(out, out_s)
    for (pr, ch), (pr_s, ch_s) in zip(saved, saved_s):
        assert (pr is None) == (pr_s is None)
        if pr is not None:
            assert torch.equal(pr.view(torch.int32), pr_s.view(torch.int32))
        if ch is not None:
            assert torch.equal(ch, ch_s)
    if last:       # leaves on one parent hold identical rows: the chirality branch's raw-row equality must have fired, both ways
        eq_rows = sum(int((ch == 1).all(dim=1).sum()) for _, ch in saved if ch is not None)
        assert eq_rows > 0


@pytest.mark.parametrize("last", [False, True])
def test_backward_on_presplit_rows(last):
    """Two layers, the h between them pre-split or not: identical forward, bank gradients of the second layer bit for bit, every
    other gradient to fp32 rounding (the gather of the second layer un-normalises with x to 2^-22)."""
    from molkgnn_amd import functional as Fn
    dev, b, plan, first, second, store = _setup(28, dup=0.2)
    p1, E = first._bank_params("train", b.x)
    p2, _ = second._bank_params("train", b.x)
    cot = torch.randn(b.x.shape[0], 110, device=dev)

    def run(split):
        for p in list(p1) + list(p2):
            p.grad = None
        x = store[:, :28].detach().requires_grad_(True)
        h = Fn.kernelsetconv(x, plan, False, p1, E, "auto", block_rows=True, propagate="split" if split else True)
        out = Fn.kernelsetconv(h, plan, last, p2, E, "auto", block_rows=True, propagate=True)
        (out * cot).sum().backward()
        torch.cuda.synchronize()
        return (out.detach().clone(), x.grad.clone(), [None if p.grad is None else p.grad.clone() for p in p2],
                [None if p.grad is None else p.grad.clone() for p in p1])

    o0, gx0, g2_0, g1_0 = run(False)
    o1, gx1, g2_1, g1_1 = run(True)
    assert torch.equal(o0, o1)
    for a, c in zip(g2_0, g2_1):
        assert (a is None) == (c is None)
        if a is not None:
            assert torch.equal(a, c)
    scale = float(gx0.abs().max())
    assert float((gx0 - gx1).abs().max()) <= 2e-6 * scale, float((gx0 - gx1).abs().max()) / scale
    for a, c in zip(g1_0, g1_1):
        assert (a is None) == (c is None)
        if a is not None:
            # (sums over ~10^4 (atom, kernel) terms of either sign, of inputs that differ by 2e-7 of their scale)
            assert float((a - c).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-3)


def test_unsupported_consumers_refuse_presplit_rows():
    """Pre-split rows are only meaningful to the streamed kernels: the generic and the bf16 variants fail loudly."""
    from molkgnn_amd import _lib
    from molkgnn_amd import functional as Fn
    dev, b, plan, first, second, store = _setup(28)
    p1, E = first._bank_params("train", b.x)
    p2, _ = second._bank_params("train", b.x)
    with torch.no_grad():
        hs = Fn.kernelsetconv(store[:, :28], plan, False, p1, E, "auto", block_rows=True, propagate="split")
        for variant in ("generic", "bf16"):
            with pytest.raises(_lib.MolKGNNLibraryError):
                Fn.kernelsetconv_details(hs, plan, False, p2, E, variant)


@pytest.mark.parametrize("training", [True, False])
def test_network_with_and_without_presplit_rows(training, monkeypatch):
    """MolKGNNNet (3 layers, the benchmark's banks): the default -- h between layers pre-split -- against MKGNN_ROWS_SPLIT=0's
    form: the same embedding bit for bit, every gradient to fp32 rounding."""
    from molkgnn_amd import KernelLayer
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(3)
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    model = MolKGNNNet(num_layers=3, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32,
                       **dict(zip(names, (10, 20, 30, 50) * 2))).to(dev)
    model.train(training)
    b = make_batch(300, seed=77).to(dev)
    assert KernelLayer._ROWS_SPLIT, "pre-split rows are the default"

    def run():
        for p in model.parameters():
            p.grad = None
        state = {k: v.clone() for k, v in model.state_dict().items()}
        emb = model(b)
        if training:
            emb.square().mean().backward()
        torch.cuda.synchronize()
        model.load_state_dict(state)                      # (the running statistics moved)
        return emb.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    calls = []
    from molkgnn_amd import functional as Fn
    real = Fn._segment_sum_blocks

    def spy(v, csr, deg8, blocks, mode, out_pad, inv):
        calls.append(mode)
        return real(v, csr, deg8, blocks, mode, out_pad, inv)
    monkeypatch.setattr(Fn, "_segment_sum_blocks", spy)
    e1, g1 = run()
    assert calls.count(3) == 2, calls                     # both inner h's were written pre-split
    calls.clear()
    monkeypatch.setattr(KernelLayer, "_ROWS_SPLIT", False)
    e0, g0 = run()
    assert calls.count(3) == 0
    assert torch.equal(e0, e1)
    assert set(g0) == set(g1)
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * max(float(g0[n].abs().max()), 1e-6), n


@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_writes_presplit_rows_for_the_first_layer(training):
    """readout.batch_norm(split_out=True): the normalised features as pre-split rows (28 channels: the two-chunk kernels, whose
    bank gradient reads the swizzled slot image).  Against the fp32 rows of the same batch norm: rows decode to 2^-22, the same
    row norms, the first layer's forward bit for bit, its bank gradients bit for bit, the batch norm's own gradients to rounding."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd import readout as R
    dev, b, plan, first, _, _ = _setup(28, dup=0.0)
    p1, E = first._bank_params("train", b.x)
    assert Fn.rows_split_supported(plan, p1, 28, E, plan.n_atoms)
    torch.manual_seed(2)
    bn = torch.nn.BatchNorm1d(28).to(dev).train(training)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 2.0)
    state = {k: v.clone() for k, v in bn.state_dict().items()}
    raw = (torch.randn(b.x.shape[0], 28, device=dev) * 1.7 + 0.3)
    cot = torch.randn(b.x.shape[0], 110, device=dev)

    def run(split):
        bn.load_state_dict(state)
        for p in list(p1) + list(bn.parameters()):
            p.grad = None
        xin = raw.detach().clone().requires_grad_(True)
        x = R.batch_norm(xin, bn, split_out=split)
        assert Fn.is_rows_split(x) == split
        inv = getattr(x, Fn._INV_ATTR)[0]
        out = Fn.kernelsetconv(x, plan, False, p1, E, "auto", block_rows=True, propagate=True)
        (out * cot).sum().backward()
        torch.cuda.synchronize()
        return (x.detach().clone(), inv.clone(), out.detach().clone(), xin.grad.clone(), [None if p.grad is None else p.grad.clone() for p in p1],
                [p.grad.clone() for p in bn.parameters()], {k: v.clone() for k, v in bn.state_dict().items()})

    x0, i0, o0, gx0, g0, gb0, s0 = run(False)
    x1, i1, o1, gx1, g1, gb1, s1 = run(True)
    assert torch.equal(i0, i1) and torch.equal(o0, o1)
    back = _decode(x1, i1, 28)
    e = (i1.view(torch.int32) >> 23) & 0xFF
    scale = torch.exp2((e - 127 + 8).float())[:, None]
    assert bool(((back - x0).abs() * scale <= 2.0 ** -22 * x0.abs() * scale + 2.0 ** -25).all())
    for a, c in zip(g0, g1):
        assert (a is None) == (c is None)
        if a is not None:
            assert torch.equal(a, c)
    assert float((gx0 - gx1).abs().max()) <= 2e-6 * float(gx0.abs().max())
    for a, c in zip(gb0, gb1):
        assert float((a - c).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-6)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
