"""The fused tail of a training step (round 6; csrc/kgnn_tail.hip, readout.tail_loss): propagate -> readout -> head -> loss and
every gradient behind the last kernel convolution in one launch (reference KernelLayer.py:119-123, MolKGNNNet.py:144-146,
model.py:147-150, 169, 190-198).  ``pytest -m gpu``.

Checked against the separate operators it replaces -- ``readout_blocks`` + ``bce_head_loss``, which tests/test_hip_parity.py pins
to the PyTorch formulas and to the reference's golden network -- and against the PyTorch formula directly.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    return torch.device("cuda:0")


def _block_rows(batch, plan, Ls, dev, seed):
    """Random block rows [N, 112] (NaN outside every atom's own block: nothing may read there) and the mask of the blocks."""
    g = torch.Generator(device=dev).manual_seed(seed)
    n, K = batch.x.shape[0], sum(Ls)
    store = torch.full((n, K + (-K) % 4), float("nan"), device=dev)
    inblock = torch.zeros(n, K, dtype=torch.bool, device=dev)
    off = 0
    for d, L in enumerate(Ls, start=1):
        sel = plan.buckets[d - 1].sel
        if sel.numel():
            inblock[sel, off:off + L] = True
        off += L
    vals = torch.randn(n, K, generator=g, device=dev)
    store[:, :K] = torch.where(inblock, vals, torch.full_like(vals, float("nan")))
    return store[:, :K], inblock


def _modules(dev, K=110, H=32, G=32):
    torch.manual_seed(5)
    lin1 = torch.nn.Linear(K, H).to(dev)
    lin2 = torch.nn.Linear(H, G).to(dev)
    ffn = torch.nn.Linear(G, 1).to(dev)
    return lin1, lin2, ffn


@pytest.mark.parametrize("p_drop,n_pad", [(0.0, 0), (0.25, 0), (0.0, 7), (0.25, 7)])
def test_fused_tail_matches_the_separate_operators(p_drop, n_pad):
    from molkgnn_amd import readout as R
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    Ls = (10, 20, 30, 50)
    b = make_batch(300, seed=41).to(dev)
    plan = plan_from_data(b)
    seg = R.molecule_segments(b.batch, 300)
    assert R.tail_supported(110, 32, 32, Ls) and R._tail_limits_ok(seg, plan)
    lin1, lin2, ffn = _modules(dev)
    sim0, inblock = _block_rows(b, plan, Ls, dev, 3)
    n_rows = 300 - n_pad                                  # the last molecules: padding, outside the loss
    y = (torch.rand(n_rows, device=dev) < 0.3).float()
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(ffn.parameters())

    def run(fused):
        R.reset_head_rng(dev, seed=1234)
        for p in params:
            p.grad = None
        sim = sim0.detach().requires_grad_(True)
        if fused:
            loss = R.tail_loss(sim, plan, Ls, lin1, lin2, ffn, y, seg, p_drop, n_rows)
        else:
            emb = R.readout_blocks(sim, plan, Ls, lin1, lin2, None, seg)
            loss = R.bce_head_loss(emb, ffn, y, dropout_p=p_drop, n_rows=n_rows)
        loss.backward()
        torch.cuda.synchronize()
        state = R.head_rng_state(dev).clone() if p_drop > 0 else None
        return loss.detach().clone(), torch.where(inblock, sim.grad, torch.zeros((), device=dev)), [p.grad.clone() for p in params], state

    l0, gs0, gp0, st0 = run(False)
    l1, gs1, gp1, st1 = run(True)
    assert torch.isfinite(l1) and abs(float(l0) - float(l1)) <= 2e-6 * max(1.0, abs(float(l0))), (float(l0), float(l1))
    assert float((gs0 - gs1).abs().max()) <= 2e-5 * max(float(gs0.abs().max()), 1e-8)
    for a, c, nm in zip(gp0, gp1, ("w1", "b1", "w2", "b2", "wh", "bh")):
        assert float((a - c).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-6), nm
    if p_drop > 0:
        assert torch.equal(st0, st1)                      # the generator advanced the same way


def test_fused_tail_against_the_pytorch_formula():
    """No dropout: loss and every gradient against autograd of the reference's formula on the dense h = propagate(sim)."""
    from molkgnn_amd import readout as R
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    Ls = (10, 20, 30, 50)
    b = make_batch(200, seed=43).to(dev)
    plan = plan_from_data(b)
    seg = R.molecule_segments(b.batch, 200)
    lin1, lin2, ffn = _modules(dev)
    sim0, inblock = _block_rows(b, plan, Ls, dev, 9)
    y = (torch.rand(200, device=dev) < 0.3).float()
    params = list(lin1.parameters()) + list(lin2.parameters()) + list(ffn.parameters())
    sim = sim0.detach().requires_grad_(True)
    loss = R.tail_loss(sim, plan, Ls, lin1, lin2, ffn, y, seg, 0.0, None)
    loss.backward()
    got = [loss.detach().double()] + [torch.where(inblock, sim.grad, torch.zeros((), device=dev)).double()] + [p.grad.double() for p in params]
    for p in params:
        p.grad = None
    dense = torch.where(inblock, sim0, torch.zeros((), device=dev)).double().requires_grad_(True)
    src, dst = b.edge_index[0], b.edge_index[1]
    h = torch.zeros_like(dense).index_add_(0, dst, dense[src])                       # KernelLayer.py:119-123
    z = h @ lin1.weight.double().t() + lin1.bias.double()
    z = z * torch.sigmoid(z)
    z = z @ lin2.weight.double().t() + lin2.bias.double()
    emb = torch.zeros(200, 32, dtype=torch.float64, device=dev).index_add_(0, b.batch, z)     # MolKGNNNet.py:144-146
    pred = emb @ ffn.weight.double().t() + ffn.bias.double()
    ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(pred.view(-1), y.double())
    grads = torch.autograd.grad(ref_loss, [dense] + [p for p in params])
    want = [ref_loss.detach()] + [torch.where(inblock, grads[0], torch.zeros((), device=dev, dtype=torch.float64))] + [g.double() for g in grads[1:]]
    for g, w, nm in zip(got, want, ("loss", "gsim", "w1", "b1", "w2", "b2", "wh", "bh")):
        assert float((g - w).abs().max()) <= 2e-5 * max(float(w.abs().max()), 1e-6), (nm, float((g - w).abs().max()), float(w.abs().max()))


def test_model_loss_takes_the_fused_tail_and_gives_the_same_step(monkeypatch):
    """GNNModel.loss on a batch above the one-launch molecule path: the fused tail is taken (spied), and loss and every parameter
    gradient agree with MKGNN_FUSED_TAIL=0's separate operators; a batch with one molecule beyond a chunk falls back."""
    from molkgnn_amd import readout as R
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, backward as train_backward
    dev = _dev()
    torch.manual_seed(1798)
    model = GNNModel(ffn_dropout_rate=0.25).to(dev)
    model.train()
    b = make_batch(600, seed=47).to(dev)
    calls = []
    real = R.tail_loss

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    monkeypatch.setattr(R, "tail_loss", spy)

    def step(fused):
        monkeypatch.setattr(R, "_FUSED_TAIL", fused)
        R.reset_head_rng(dev, seed=99)
        state = {k: v.clone() for k, v in model.state_dict().items()}
        model.zero_grad(set_to_none=True)
        loss = model.loss(b)
        train_backward(loss)
        torch.cuda.synchronize()
        model.load_state_dict(state)
        return float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    l1, g1 = step(True)
    assert len(calls) == 1
    l0, g0 = step(False)
    assert len(calls) == 1
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)), (l0, l1)
    assert set(g0) == set(g1) and len(g0) > 70
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 5e-5 * max(float(g0[n].abs().max()), 1e-6), n


def test_oversize_molecule_is_refused_by_the_host_and_loud_in_the_kernel():
    from molkgnn_amd import readout as R
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    Ls = (10, 20, 30, 50)
    b = make_batch(40, seed=51).to(dev)
    plan = plan_from_data(b)
    # the molecule vector says: the first 200 atoms are ONE molecule (beyond MKGNN_TAIL_MAX_ATOMS)
    batch = b.batch.clone()
    first = int((batch < 9).sum())
    assert first > 128
    merged = torch.where(batch < 9, torch.zeros_like(batch), batch - 8)
    seg = R.MoleculeSegments(merged, int(merged.max()) + 1)
    assert not R._tail_limits_ok(seg, plan)
    lin1, lin2, ffn = _modules(dev)
    sim0, _ = _block_rows(b, plan, Ls, dev, 2)
    y = torch.zeros(seg.size, device=dev)
    loss = R.tail_loss(sim0.detach().requires_grad_(True), plan, Ls, lin1, lin2, ffn, y, seg, 0.0, None)
    assert torch.isnan(loss)
    # ... and the status word is clean again: an ordinary batch right after gives a finite loss
    seg_ok = R.molecule_segments(b.batch, 40)
    loss = R.tail_loss(sim0.detach().requires_grad_(True), plan, Ls, lin1, lin2, ffn, torch.zeros(40, device=dev), seg_ok, 0.0, None)
    assert torch.isfinite(loss)


def test_fused_tail_is_padding_invariant():
    """A batch padded to a fixed shape (padding.pad_batch: 64 padding molecules behind the real ones, outside the loss) through
    GNNModel.loss with the fused tail: the loss is BIT FOR BIT the unpadded batch's -- molecules are dealt to workgroups in groups
    whose composition depends on the number of real molecules only, and the padding molecules add exact zeros -- and every
    gradient agrees to rounding (the padded degree buckets are cut into tiles differently)."""
    from molkgnn_amd import padding as P
    from molkgnn_amd import readout as R
    from molkgnn_amd.receptive_field import attach_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    dev = _dev()
    torch.manual_seed(3)
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    B = 700
    raw = make_batch(B, seed=4100, with_receptive_fields=False)
    raw.y = (torch.arange(B) % 5 == 0).long()
    other = make_batch(B, seed=4101, with_receptive_fields=False)
    shape = P.fixed_shape([P.degree_histogram(raw), P.degree_histogram(other)])
    calls = []
    real = R.tail_loss

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    R.tail_loss = spy
    try:
        res = []
        for batch in (attach_receptive_fields(raw.to(dev)),
                      attach_receptive_fields(P.pad_batch(raw, shape, B).to(dev), sizes=[shape[f"n{d}"] for d in range(1, 5)])):
            state = {k: v.clone() for k, v in model.state_dict().items()}
            model.zero_grad(set_to_none=True)
            loss = model.loss(batch)
            loss.backward()
            torch.cuda.synchronize()
            model.load_state_dict(state)
            res.append((loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    finally:
        R.tail_loss = real
    assert len(calls) == 2                                # both through the fused tail (pad_batch hands the molecule bound over)
    assert torch.equal(res[0][0], res[1][0]), (float(res[0][0]), float(res[1][0]))
    for n, g in res[0][1].items():
        assert float((g - res[1][1][n]).abs().max()) <= 2e-5 * max(float(g.abs().max()), 1e-3) + 1e-7, n


def test_deferred_tail_reduction_gives_the_same_step(monkeypatch):
    """``train.training_step`` / ``readout.deferred_tail_reduce`` (round 6): inside the region the fused tail's last launch -- the
    reduction that writes the loss and the six parameter gradients -- is made by the first kernel convolution backward on its helper
    stream (spied: ``mkgnn_tail_flush`` at the region's end finds nothing left), and the step is bit for bit the three separate calls:
    loss, every gradient, every parameter after AdamW.  A backward that must READ those gradients early -- a seed that is not the
    registered one, a ``.grad`` that already exists -- flushes first and is still exact."""
    from molkgnn_amd import readout as R
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, backward as train_backward, configure_optimizer, training_step
    dev = _dev()
    b = make_batch(600, seed=47).to(dev)

    def fresh():
        torch.manual_seed(1798)
        model = GNNModel(ffn_dropout_rate=0.25).to(dev)
        model.train()
        R.reset_head_rng(dev, seed=99)
        return model, configure_optimizer(model, lr=1e-3, capturable=True)

    # (a) the three calls
    model, opt = fresh()
    model.zero_grad(set_to_none=True)
    la = model.loss(b)
    train_backward(la)
    ga = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    pa = {n: p.detach().clone() for n, p in model.named_parameters()}
    # (b) one unit
    flushed = []
    real_flush = R.tail_flush

    def spy_flush(device):
        flushed.append(1)
        return real_flush(device)
    monkeypatch.setattr(R, "tail_flush", spy_flush)
    model, opt = fresh()
    lb = training_step(model, b, opt)
    torch.cuda.synchronize()
    assert flushed, "the region flushes at its end"
    assert float(lb) == float(la)
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), pa[n]), n
    # (c) gradients themselves, and the two early-read cases
    model, opt = fresh()
    model.zero_grad(set_to_none=True)
    with R.deferred_tail_reduce(dev):
        lc = model.loss(b)
        train_backward(lc)
    torch.cuda.synchronize()
    gc = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert gc.keys() == ga.keys() and all(torch.equal(gc[n], ga[n]) for n in ga)
    model, opt = fresh()
    model.zero_grad(set_to_none=True)
    with R.deferred_tail_reduce(dev):
        ld = model.loss(b)
        (2.0 * ld).backward()                            # not the registered unit seed: the tail's backward scales its gradients
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        if n in ga:
            assert torch.allclose(p.grad, 2.0 * ga[n], rtol=2e-5, atol=1e-7 * float(ga[n].abs().max())), n
    model, opt = fresh()
    for p in model.parameters():
        p.grad = torch.ones_like(p)                      # existing gradients: autograd accumulates into them
    with R.deferred_tail_reduce(dev):
        le = model.loss(b)
        le.backward()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        if n in ga:
            assert torch.allclose(p.grad, 1.0 + ga[n], rtol=1e-6, atol=1e-6), n
