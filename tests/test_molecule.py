"""The molecule-resident small-batch path (molkgnn_amd.molecule, csrc/kgnn_molecule.hip) against the ORACLE.  ``pytest -m gpu``.

Same criteria as the per-operator path: every layer's scores by the tie-aware criterion against the
reference-faithful oracle form, the network's embedding / loss and every parameter gradient against the oracle evaluated
with the build's own permutation choices -- at 1, 2, 16 and 256 molecules, with absent degrees, a molecule of more than 32
atoms (the full-size kernel variant), several molecules per chunk, training- and eval-mode batch norm.
"""
import pytest
import torch

from oracle import kgnn_oracle as O

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    return torch.device("cuda:0")


def _model(counts, layers, hidden, seed, dev, train_bn):
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    torch.manual_seed(seed)
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    model = MolKGNNNet(num_layers=layers, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=hidden,
                       **dict(zip(names, counts * 2)))
    with torch.no_grad():
        model.node_batch_norm.running_mean.normal_(0.0, 0.3)
        model.node_batch_norm.running_var.uniform_(0.5, 1.5)
        model.node_batch_norm.weight.uniform_(0.5, 1.5)
        model.node_batch_norm.bias.normal_(0.0, 0.2)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    model.train(train_bn)
    return model, state


def _forced_from_capture(cap, layers):
    forced = []
    for li in range(layers):
        idx = []
        for d in range(4):
            sv = cap["saved"][li][d]
            idx.append(None if sv is None else sv[0][..., 3].contiguous().view(torch.int32).t().cpu().long())
        forced.append(idx)
    return forced


def _check_against_oracle(model, state, b, layers, train_bn, emb, cap, cot, grads_of):
    """Layer by layer (tie-aware) and end to end (embedding, every parameter gradient) against the oracle replayed with the
    build's permutation choices."""
    forced = _forced_from_capture(cap, layers)
    ostate = {k: v.clone() for k, v in state.items()}
    h_o = O.batch_norm(b.x, ostate["node_batch_norm.weight"], ostate["node_batch_norm.bias"],
                       ostate["node_batch_norm.running_mean"].clone(), ostate["node_batch_norm.running_var"].clone(), train_bn)
    for i in range(layers):
        per_degree = O.kernelset_params(ostate, f"gnn.layers.{i}.")
        sim = cap["sims"][i].cpu()
        assert O.kernelset_tie_aware_mismatch(per_degree, h_o, b, i == layers - 1, sim, forced[i]) == 0, f"layer {i}"
        sim_o = O.kernelsetconv(per_degree, h_o, b, i == layers - 1, form="faithful", forced_idx=forced[i])
        assert torch.allclose(sim, sim_o, atol=FWD_TOL, rtol=0), (i, float((sim - sim_o).abs().max()))
        h_o = O.propagate_add(b.edge_index, sim_o)
    ostate = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in ostate.items()}
    if "node_batch_norm.running_mean" in ostate:
        ostate["node_batch_norm.running_mean"] = ostate["node_batch_norm.running_mean"].clone()
        ostate["node_batch_norm.running_var"] = ostate["node_batch_norm.running_var"].clone()
    emb_o = O.molkgnnnet(ostate, b, layers, training_bn=train_bn, form="faithful", forced_idx=forced)
    scale = max(1.0, float(emb_o.detach().abs().max()))
    assert float((emb.detach().cpu() - emb_o.detach()).abs().max()) <= 5e-5 * scale
    if cot is None:
        return ostate, emb_o
    (emb_o * cot).sum().backward()
    checked = 0
    for nm, got in grads_of.items():
        ref = ostate[nm].grad
        if got is None:
            assert ref is None or float(ref.abs().max()) == 0.0, nm
            continue
        assert ref is not None, nm
        err = float((got.cpu() - ref).abs().max())
        assert err <= 5e-5 * max(1.0, float(ref.abs().max())) + 1e-3 * float(ref.abs().max()), (nm, err, float(ref.abs().max()))
        checked += 1
    return checked


@pytest.mark.parametrize("mols,train_bn,counts,layers,hidden", [
    (16, False, (10, 20, 30, 50), 3, 32), (16, True, (10, 20, 30, 50), 3, 32), (1, False, (10, 20, 30, 50), 3, 32),
    (2, True, (10, 20, 30, 50), 3, 32), (40, True, (5, 10, 15, 25), 3, 32), (24, False, (1, 1, 1, 1), 4, 32),
    (256, True, (10, 20, 30, 50), 3, 32), (12, True, (10, 20, 30, 50), 2, 64)])
def test_network_forward_and_backward_against_the_oracle(mols, train_bn, counts, layers, hidden, monkeypatch):
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    monkeypatch.setattr(M, "_MODE", "1")
    model, state = _model(counts, layers, hidden, sum(counts) + layers + mols, dev, train_bn)
    b = make_batch(mols, seed=sum(counts) + mols, duplicate_fraction=0.1)
    b.num_graphs = mols
    bd = b.to(dev)
    cot = torch.randn(mols, hidden, generator=torch.Generator().manual_seed(1))
    cap = {}
    monkeypatch.setattr(M, "debug_capture", cap)
    calls = []
    orig = M._run
    monkeypatch.setattr(M, "_run", lambda *a, **k: (calls.append(a[6]), orig(*a, **k))[1])
    emb = model(bd)
    assert calls == [0], calls                             # the molecule-resident kernels ran (forward-only mode)
    fwd_cap = dict(cap)
    (emb * cot.to(dev)).sum().backward()
    assert calls == [0, 6], calls                          # ... and their backward (recompute + gradient of the embedding)
    grads = {nm: prm.grad for nm, prm in model.named_parameters()}
    checked = _check_against_oracle(model, state, b, layers, train_bn, emb, fwd_cap, cot, grads)
    present = sum(1 for d in range(1, 5) if getattr(b, f"selected_index_deg{d}").numel() > 0)
    assert checked >= 6 * present * layers + 6
    if train_bn:                                            # running statistics moved once (the recompute does not move them again)
        rm = torch.nn.functional.batch_norm(b.x, state["node_batch_norm.running_mean"].clone(), state["node_batch_norm.running_var"].clone(),
                                            None, None, True, 0.1, 1e-5)
        ref_m, ref_v = state["node_batch_norm.running_mean"].clone(), state["node_batch_norm.running_var"].clone()
        torch.nn.functional.batch_norm(b.x, ref_m, ref_v, None, None, True, 0.1, 1e-5)
        assert torch.allclose(model.node_batch_norm.running_mean.cpu(), ref_m, atol=1e-5)
        assert torch.allclose(model.node_batch_norm.running_var.cpu(), ref_v, atol=1e-5, rtol=1e-5)
        assert int(model.node_batch_norm.num_batches_tracked) == 1
        del rm


def _big_molecule_batch(dev):
    """Molecules of up to 60 atoms next to small ones: chunks of more than 32 atoms (the full-size kernel variant), several
    molecules per chunk, and a chunk table that mixes both."""
    from molkgnn_amd.synthetic import make_batch
    for seed in range(200, 400):
        b = make_batch(24, seed=seed, duplicate_fraction=0.1)
        sizes = torch.bincount(b.batch).tolist()
        if max(sizes) >= 40 and min(sizes) <= 14:
            b.num_graphs = 24
            return b
    raise AssertionError("no batch with a large and a small molecule found")


def test_large_molecules_and_packed_chunks(monkeypatch):
    dev = _dev()
    from molkgnn_amd import molecule as M
    monkeypatch.setattr(M, "_MODE", "1")
    b = _big_molecule_batch(dev)
    bd = b.to(dev)
    model, state = _model((10, 20, 30, 50), 3, 32, 5, dev, True)
    cap = {}
    monkeypatch.setattr(M, "debug_capture", cap)
    plan = M._plan_of(bd)
    plan._molecule = (M.build_molecule_plan(plan, bd.batch, 24, cap=64),)          # chunks of up to 64 atoms: several molecules each
    emb = model(bd)
    mp = M.molecule_plan(plan, bd.batch, 24)
    assert mp is not None and mp.max_chunk_atoms > 48 and mp.n_chunks < 16, (mp.max_chunk_atoms, mp.n_chunks)
    fwd_cap = dict(cap)
    cot = torch.randn(24, 32, generator=torch.Generator().manual_seed(2))
    (emb * cot.to(dev)).sum().backward()
    grads = {nm: prm.grad for nm, prm in model.named_parameters()}
    assert _check_against_oracle(model, state, b, 3, True, emb, fwd_cap, cot, grads) > 60


@pytest.mark.parametrize("mols,p_drop", [(16, 0.0), (256, 0.0), (16, 0.25)])
def test_training_step_loss_and_gradients(mols, p_drop, monkeypatch):
    """``GNNModel.loss`` as ONE launch (forward + BCE head + backward): loss and every gradient against the oracle replayed
    with the build's choices (dropout 0), and with dropout against the per-operator head fed the same embedding and the
    same generator state."""
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd import readout as R
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    from molkgnn_amd.train import backward as train_backward
    monkeypatch.setattr(M, "_MODE", "1")
    torch.manual_seed(40 + mols)
    model = GNNModel(ffn_dropout_rate=p_drop)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    b = make_batch(mols, seed=900 + mols, duplicate_fraction=0.1)
    b.num_graphs = mols
    b.y = (torch.rand(mols, generator=torch.Generator().manual_seed(3)) < 0.3).float()
    bd = b.to(dev)
    cap = {}
    monkeypatch.setattr(M, "debug_capture", cap)
    calls = []
    orig = M._run
    monkeypatch.setattr(M, "_run", lambda *a, **k: (calls.append(a[6]), orig(*a, **k))[1])
    R.reset_head_rng(dev, seed=77)
    loss = model.loss(bd)
    assert calls == [3], calls                              # HEAD | BACKWARD in one launch
    train_backward(loss)
    assert calls == [3]                                     # the backward launched nothing
    gstate = {k[len("gnn_model."):]: v for k, v in state.items() if k.startswith("gnn_model.")}
    grads = {nm[len("gnn_model."):]: prm.grad for nm, prm in model.named_parameters() if nm.startswith("gnn_model.")}
    if p_drop == 0.0:
        forced = _forced_from_capture(cap, 3)
        ostate = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in gstate.items()}
        emb_o = O.molkgnnnet(ostate, b, 3, training_bn=True, form="faithful", forced_idx=forced)
        w = state["ffn.weight"].clone().requires_grad_(True)
        bias = state["ffn.bias"].clone().requires_grad_(True)
        pred_o = emb_o @ w.T + bias
        loss_o = torch.nn.functional.binary_cross_entropy_with_logits(pred_o.view(-1), b.y)
        assert abs(float(loss) - float(loss_o)) <= 2e-6 * max(1.0, abs(float(loss_o)))
        assert torch.allclose(cap["pred"].cpu(), pred_o.detach().view(-1), atol=5e-5, rtol=1e-5)
        loss_o.backward()
        checked = 0
        for nm, got in list(grads.items()) + [("ffn.weight", model.ffn.weight.grad), ("ffn.bias", model.ffn.bias.grad)]:
            ref = {"ffn.weight": w.grad, "ffn.bias": bias.grad}.get(nm, ostate[nm].grad if nm in ostate else None)
            if got is None:
                assert ref is None or float(ref.abs().max()) == 0.0, nm
                continue
            err = float((got.cpu() - ref).abs().max())
            assert err <= 2e-6 * max(1.0, float(ref.abs().max())) + 1e-3 * float(ref.abs().max()), (nm, err, float(ref.abs().max()))
            checked += 1
        assert checked >= 70
    else:
        # the per-operator head on the SAME embedding with the same generator state draws the same mask
        emb = cap["emb"].detach().clone().requires_grad_(True)
        w = model.ffn.weight.detach().clone().requires_grad_(True)
        bias = model.ffn.bias.detach().clone().requires_grad_(True)
        ffn = torch.nn.Linear(32, 1).to(dev)
        ffn.weight, ffn.bias = torch.nn.Parameter(w), torch.nn.Parameter(bias)
        R.reset_head_rng(dev, seed=77)
        loss2 = R.bce_head_loss(emb, ffn, bd.y, dropout_p=p_drop)
        loss2.backward()
        assert abs(float(loss) - float(loss2)) <= 2e-6
        assert torch.allclose(model.ffn.weight.grad, ffn.weight.grad, atol=2e-6, rtol=1e-5)
        assert torch.allclose(model.ffn.bias.grad, ffn.bias.grad, atol=2e-6, rtol=1e-5)
        state_now = R.head_rng_state(dev).tolist()
        assert state_now[1] == 1                            # (reset to offset 0, one draw by the per-operator head after the reset)


def test_step_under_graph_capture_replays_the_eager_step(monkeypatch):
    """forward + loss + backward + AdamW captured once and replayed: same loss and parameters as the eager steps."""
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, configure_optimizer
    from molkgnn_amd.train import backward as train_backward
    monkeypatch.setattr(M, "_MODE", "1")
    b = make_batch(16, seed=31).to(dev)

    def run(graph):
        torch.manual_seed(8)
        model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
        opt = configure_optimizer(model, lr=1e-3, capturable=True)

        def step():
            model.zero_grad(set_to_none=True)
            loss = model.loss(b)
            train_backward(loss)
            opt.step()
            return loss
        losses = []
        if not graph:
            for _ in range(5):
                losses.append(float(step()))
        else:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    losses.append(float(step()))
                model.zero_grad(set_to_none=True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    sl = step()
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3):
                g.replay()
                torch.cuda.synchronize()
                losses.append(float(sl))
        return losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0[0] > 0 and all(abs(a - c) <= 1e-6 * max(1.0, abs(a)) for a, c in zip(l0, l1)), (l0, l1)
    assert torch.allclose(p0, p1, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("mols", [16, 200])
def test_step_is_bit_reproducible(mols, monkeypatch):
    """Every sum on the path has a fixed order (owner threads for the coefficient scatter, chunk-ascending slab sums, no
    float atomics): the same step run five times gives the same bits -- loss, every gradient, the pair records.  (A missing
    barrier between two LDS phases shows up here long before it shows up against the oracle.)"""
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    from molkgnn_amd.train import backward as train_backward
    monkeypatch.setattr(M, "_MODE", "1")
    torch.manual_seed(5)
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    b = make_batch(mols, seed=77 + mols, duplicate_fraction=0.1)
    b.num_graphs = mols
    bd = b.to(dev)
    runs = []
    for _ in range(5):
        cap = {}
        monkeypatch.setattr(M, "debug_capture", cap)
        model.zero_grad(set_to_none=True)
        loss = model.loss(bd)
        train_backward(loss)
        torch.cuda.synchronize()
        grads = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
        pairs = [sv[0].clone() for layer in cap["saved"] for sv in layer if sv is not None]
        runs.append((loss.detach().clone(), grads, pairs))
    for loss, grads, pairs in runs[1:]:
        assert torch.equal(loss, runs[0][0])
        assert all((a is None and c is None) or torch.equal(a, c) for a, c in zip(grads, runs[0][1]))
        assert all(torch.equal(a, c) for a, c in zip(pairs, runs[0][2]))
    assert sum(g is not None for g in runs[0][1]) >= 70


def test_zero_norm_rows_through_the_one_launch_step(monkeypatch):
    """SURVEY 8 a-6 in the molecule-resident kernels: a row of norm < 1e-8 is divided by 1e-8, not by its norm (its cosines
    are 0, its gradient is v_hat / eps-sized and takes no projection).  With the batch norm's weight and bias at zero EVERY
    first-layer input row is exactly zero; half of the atoms get a bias back so that both branches run in one chunk."""
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    monkeypatch.setattr(M, "_MODE", "1")
    model, state = _model((10, 20, 30, 50), 3, 32, 11, dev, False)
    with torch.no_grad():
        model.node_batch_norm.weight.zero_()
        model.node_batch_norm.bias.zero_()
        # columns 0..13: a feature that is non-zero only for every other atom survives through the running statistics
        model.node_batch_norm.weight[:14] = 1.0
        model.node_batch_norm.running_mean[:14] = 0.0        # (x = 0 in the live columns is then normalised to exactly 0)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    b = make_batch(6, seed=5)
    b.num_graphs = 6
    with torch.no_grad():
        x = b.x.clone()
        rm, rv = state["node_batch_norm.running_mean"], state["node_batch_norm.running_var"]
        x[:, :14] = 0.0                                      # (x - mean) = 0 in the live columns ...
        x[1::2, :14] += torch.randn(x[1::2, :14].shape, generator=torch.Generator().manual_seed(2)) * rv[:14].sqrt()   # ... except for every other atom
        b.x = x
    bd = b.to(dev)
    cap = {}
    monkeypatch.setattr(M, "debug_capture", cap)
    emb = model(bd)
    fwd_cap = dict(cap)
    cot = torch.randn(6, 32, generator=torch.Generator().manual_seed(4))
    (emb * cot.to(dev)).sum().backward()
    x0 = torch.nn.functional.batch_norm(b.x, rm.clone(), rv.clone(), state["node_batch_norm.weight"], state["node_batch_norm.bias"], False, 0.1, 1e-5)
    assert int((x0.norm(dim=1) == 0).sum()) >= b.x.shape[0] // 2 - 1 and int((x0.norm(dim=1) > 1e-3).sum()) >= b.x.shape[0] // 2 - 1
    grads = {nm: prm.grad for nm, prm in model.named_parameters()}
    # (gradients through 1 / eps are 1e7-sized: the relative part of the criterion carries them)
    assert _check_against_oracle(model, state, b, 3, False, emb, fwd_cap, cot, grads) > 60


@pytest.mark.parametrize("mols", [24, 64, 256])
def test_default_mode_eager_warm_up_then_capture(mols, monkeypatch):
    """with MKGNN_MOLECULE unset an eager step of 33 .. 512 molecules takes the one-launch path while the
    same step inside a hipGraph capture takes the per-operator kernels -- so the eager warm-up never ran the per-operator path
    for that batch and the CAPTURED step was its first use, lazy per-batch builds (a host synchronisation in
    readout.MoleculeSegments) included.  The eager steps now build those caches; warm-up + capture + replay must work in the
    default mode at every size, and replay what the eager step of the captured path computes."""
    dev = _dev()
    import copy
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, configure_optimizer
    from molkgnn_amd.train import backward as train_backward
    monkeypatch.setattr(M, "_MODE", "")                      # the default
    torch.manual_seed(9)
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    twin = copy.deepcopy(model)
    opt, opt_t = configure_optimizer(model, lr=1e-3, fused=True), configure_optimizer(twin, lr=1e-3, fused=True)
    b = make_batch(mols, seed=1234 + mols).to(dev)
    b.num_graphs = mols
    b.y = (torch.arange(mols, device=dev) % 4 == 0).long()
    calls = []
    orig = M._run
    monkeypatch.setattr(M, "_run", lambda *a, **k: (calls.append(a[6]), orig(*a, **k))[1])

    def step(m, o):
        m.zero_grad(set_to_none=True)
        loss = m.loss(b)
        train_backward(loss)
        o.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                                   # eager warm-up: up to 32 molecules the one-launch step; above, per operator
            step(model, opt)                                 # on every visit (no ``resident`` mark on the batch: round 6)
        assert len(calls) == (2 if mols <= 32 else 0), calls
        model.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):               # captured: one launch up to 32 molecules, per operator above
            static_loss = step(model, opt)
    torch.cuda.current_stream().wait_stream(side)
    assert len(calls) == (3 if mols <= 32 else 0), calls
    g.replay()
    torch.cuda.synchronize()
    # the twin: the same three steps eagerly, each forced onto the path the model took
    for mode in (("1", "1") if mols <= 32 else ("0", "0")):
        monkeypatch.setattr(M, "_MODE", mode)
        step(twin, opt_t)
    monkeypatch.setattr(M, "_MODE", "1" if mols <= 32 else "0")
    want = step(twin, opt_t)
    torch.cuda.synchronize()
    assert torch.isfinite(static_loss).all()
    assert float((static_loss.detach() - want.detach()).abs()) <= 1e-6 * max(1.0, float(want.detach().abs())), (float(static_loss.detach()), float(want.detach()))
    for (nm, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.allclose(p, q, atol=1e-6, rtol=1e-5), nm
    for (nm, p), (_, q) in zip(model.named_buffers(), twin.named_buffers()):
        assert torch.allclose(p.float(), q.float(), atol=1e-6, rtol=1e-5), nm


@pytest.mark.parametrize("mols", [16, 200])
def test_edge_batch_norm_buffers_move_in_the_one_launch_step(mols, monkeypatch):
    """Reference MolKGNNNet.py:116: ``edge_batch_norm(data.edge_attr)`` runs in every forward; the one-launch step launches its
    side effect (``readout.update_running_stats``) next to it.  Three training steps against torch.nn.BatchNorm1d."""
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    monkeypatch.setattr(M, "_MODE", "1")
    torch.manual_seed(2)
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    ref = torch.nn.BatchNorm1d(7).to(dev)
    calls = []
    orig = M._run
    monkeypatch.setattr(M, "_run", lambda *a, **k: (calls.append(a[6]), orig(*a, **k))[1])
    for s in range(3):
        b = make_batch(mols, seed=40 + s).to(dev)
        b.num_graphs = mols
        b.y = (torch.arange(mols, device=dev) % 2).long()
        model.zero_grad(set_to_none=True)
        model.loss(b).backward()
        ref(b.edge_attr)
    assert len(calls) == 3
    bn = model.gnn_model.edge_batch_norm
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=2e-6, rtol=1e-5)
    assert torch.allclose(bn.running_var, ref.running_var, atol=1e-5, rtol=2e-5)
    assert int(bn.num_batches_tracked) == 3 and int(model.gnn_model.node_batch_norm.num_batches_tracked) == 3


def test_resident_mark_selects_the_one_launch_step_from_the_first_visit(monkeypatch):
    """The eager dispatch above 32 molecules is a function of the batch alone: ``data.resident = True`` takes the
    one-launch molecule step on EVERY visit, no mark takes the per-operator kernels on every visit -- the same kernels in epoch 1
    and epoch 2 either way."""
    dev = _dev()
    from molkgnn_amd import molecule as M
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    from molkgnn_amd.train import backward as train_backward
    monkeypatch.setattr(M, "_MODE", "")
    torch.manual_seed(9)
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    calls = []
    orig = M._run
    monkeypatch.setattr(M, "_run", lambda *a, **k: (calls.append(a[6]), orig(*a, **k))[1])
    for resident, want in ((False, 0), (True, 3)):
        b = make_batch(96, seed=77).to(dev)
        b.num_graphs = 96
        b.y = (torch.arange(96, device=dev) % 4 == 0).long()
        if resident:
            b.resident = True
        calls.clear()
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            train_backward(model.loss(b))
        torch.cuda.synchronize()
        assert len(calls) == want, (resident, calls)


def test_captured_steps_replay_what_the_eager_loop_computes():
    """train.CapturedSteps: the two-line opt-in for loops that revisit their batches -- eager for the first two
    visits of a batch, a captured graph from the third on; after three epochs over four batches the parameters are those of the
    plain eager loop (same kernels, same order)."""
    dev = _dev()
    import copy
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import CapturedSteps, GNNModel, backward as train_backward, configure_optimizer
    torch.manual_seed(21)
    model = GNNModel(ffn_dropout_rate=0.0).to(dev).train()
    twin = copy.deepcopy(model)
    opt, opt_t = configure_optimizer(model, lr=1e-3, fused=True), configure_optimizer(twin, lr=1e-3, fused=True)
    batches = []
    for i in range(4):
        b = make_batch(96, seed=300 + i).to(dev)
        b.y = (torch.arange(96, device=dev) % 3 == 0).long()
        batches.append(b)
    steps = CapturedSteps(model, opt)
    losses = []
    for epoch in range(4):
        for b in batches:
            losses.append(float(steps(b)))
            twin.zero_grad(set_to_none=True)
            lt = twin.loss(b)
            train_backward(lt)
            opt_t.step()
            assert abs(losses[-1] - float(lt.detach())) <= 1e-6 * max(1.0, abs(float(lt.detach()))), (epoch, losses[-1], float(lt.detach()))
    torch.cuda.synchronize()
    assert len(steps._graphs) == 4
    for (nm, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.allclose(p, q, atol=1e-6, rtol=1e-5), nm
