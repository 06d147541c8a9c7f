"""Parity of the HIP path (through the C ABI) with the oracle and the reference's
golden vectors.  Needs an MI355X: ``pytest -m gpu``.

Tolerances: forward 1e-5 absolute (BASELINE.json north_star), tie-aware where
neighbour rows tie; gradients 2e-5 absolute + 1e-4 relative.
"""
import os

import numpy as np
import pytest
import torch

from oracle import kgnn_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5
VARIANTS = ("generic", "mfma")


def _dev():
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    return torch.device("cuda:0")


def _conv_from_case(case, dev):
    from molkgnn_amd.kernels import KernelConv
    from molkgnn_amd.receptive_field import GraphBatch
    prm = G.kc_params(case)
    init = GraphBatch(x_center=prm["x_center"].clone(), x_support=prm["x_support"].clone(),
                      edge_attr_support=prm["edge_attr_support"].clone(), p_support=prm["p_support"].clone())
    conv = KernelConv(init_kernel=init,
                      init_support_attr_sc_weight=float(prm["support_attr_sc_weight"]),
                      init_center_attr_sc_weight=float(prm["center_attr_sc_weight"]),
                      init_edge_attr_support_sc_weight=float(prm["edge_attr_support_sc_weight"]))
    return conv.to(dev)


def _run_case(case, dev, variant, tie_free):
    from molkgnn_amd import functional as Fn
    conv = _conv_from_case(case, dev)
    conv.variant = variant
    x_focal, p_focal, x_nei, p_nei, e_nei, last = G.kc_inputs(case)
    xf = x_focal.to(dev).requires_grad_(True)
    xn = x_nei.to(dev).requires_grad_(True)
    sc = conv(is_last_layer=last, x_focal=xf, p_focal=p_focal.to(dev), x_neighbor=xn, p_neighbor=p_nei.to(dev),
              edge_attr_neighbor=e_nei.to(dev))
    assert sc.shape == case["sc"].shape
    # permutation choices for the tie-aware criterion
    x_all, plan, params = conv.single_degree_problem(xf.detach(), p_focal.to(dev), xn.detach(), p_nei.to(dev), e_nei.to(dev))
    out2, saved = Fn.kernelsetconv_details(x_all, plan, last, params, e_nei.shape[-1], variant)
    d = int(case["degree"])
    idx = saved[d - 1][0].cpu()
    assert torch.equal(out2[: xf.shape[0]].T, sc.detach())          # deterministic
    bad = O.tie_aware_mismatch(sc.detach().cpu(), idx, G.kc_params(case), *G.kc_inputs(case), tol=FWD_TOL)
    assert bad == 0
    if tie_free:
        assert torch.equal(idx.long(), case["best_index"])
        assert torch.allclose(sc.detach().cpu(), case["sc"], atol=FWD_TOL, rtol=0)
        names = ["x_focal", "x_neighbor"] + list(G.PARAMS)
        tensors = [xf, xn] + [getattr(conv, k) for k in G.PARAMS]
        grads = torch.autograd.grad((sc * case["cotangent"].to(dev)).sum(), tensors, allow_unused=True)
        for nm, gr in zip(names, grads):
            if f"grad_cot_{nm}_is_none" in case:
                assert gr is None, nm
            else:
                ref = case[f"grad_cot_{nm}"]
                assert torch.allclose(gr.cpu(), ref, atol=2e-5, rtol=1e-4), (nm, float((gr.cpu() - ref).abs().max()))


@pytest.mark.parametrize("variant", VARIANTS)
def test_kernelconv_per_degree_golden(variant):
    dev = _dev()
    flat = G.load("g1_kernelconv.npz")
    for nm in G.case_names(flat):
        _run_case(G.group(flat, nm), dev, variant, tie_free=True)


@pytest.mark.parametrize("variant", VARIANTS)
def test_kernelconv_ties_golden(variant):
    dev = _dev()
    flat = G.load("g4_ties.npz")
    for nm in G.case_names(flat):
        _run_case(G.group(flat, nm), dev, variant, tie_free=False)


@pytest.mark.parametrize("variant", VARIANTS)
def test_kernelconv_chirality_golden(variant):
    dev = _dev()
    flat = G.load("g5_chirality.npz")
    _run_case(G.group(flat, "d4"), dev, variant, tie_free=True)


def _setconv_from_state(state, Ls, F, dev):
    from molkgnn_amd.kernels import KernelSetConv
    ksc = KernelSetConv(*[int(v) for v in Ls], D=3, node_attr_dim=F, edge_attr_dim=7)
    missing = ksc.load_state_dict(state, strict=True)
    return ksc.to(dev)


@pytest.mark.parametrize("variant", VARIANTS)
def test_kernelsetconv_golden(variant):
    dev = _dev()
    from molkgnn_amd.receptive_field import GraphBatch
    flat = G.load("g2_kernelsetconv.npz")
    for tag in ("all", "nodeg4"):
        b = GraphBatch(**{k[len("in_"):]: v for k, v in G.group(flat, tag).items() if k.startswith("in_")}).to(dev)
        for ltag, F in (("F28", 28), ("F110", 110)):
            sub = G.group(flat, f"{tag}_{ltag}")
            state = {k[len("param/"):]: v for k, v in sub.items() if k.startswith("param/")}
            ksc = _setconv_from_state(state, sub["L"], F, dev)
            ksc.variant = variant
            x = sub["x"].to(dev).requires_grad_(True)
            b.x = x
            for last in (False, True):
                # data mode (exactly two keyword arguments, kernels.py:622)
                sc = ksc(is_last_layer=last, data=b, save_score=False)
                assert torch.allclose(sc.detach().cpu(), sub[f"sc_last{int(last)}"], atol=FWD_TOL, rtol=0)
            # exploded keyword mode gives the same tensor
            kw = {k: getattr(b, k) for k in b.keys() if "_deg" in k}
            sc2 = ksc(is_last_layer=True, x=x, edge_index=b.edge_index, edge_attr=b.edge_attr, p=b.p, save_score=False, **kw)
            assert torch.equal(sc2.detach(), sc.detach())
            with pytest.raises(Exception):
                ksc(True, b)
            (sc * sub["cotangent"].to(dev)).sum().backward()
            assert torch.allclose(x.grad.cpu(), sub["grad_x"], atol=2e-5, rtol=1e-4)
            for nm, prm in ksc.named_parameters():
                key = f"grad/{nm}"
                if key in sub:
                    assert torch.allclose(prm.grad.cpu(), sub[key], atol=2e-5, rtol=1e-4), (nm, float((prm.grad.cpu() - sub[key]).abs().max()))
                else:
                    assert prm.grad is None, nm        # p_support, length/angle weights


def _net_from_golden(flat, dev):
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    kc = [int(v) for v in flat["kernel_counts"]]
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    model = MolKGNNNet(num_layers=3, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32,
                       **dict(zip(names, kc)))
    state = {k[len("param/"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("param/")}
    state.update({k[len("buffer/"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("buffer/")})
    model.load_state_dict(state, strict=True)
    return model.to(dev).eval(), state


@pytest.mark.parametrize("variant", VARIANTS)
def test_three_layer_network_tie_aware(variant):
    """End to end: the build's 3-layer network against the oracle evaluated with
    the build's own permutation choices, after checking that every choice is
    within 1e-6 of the oracle's maximum (layers >= 1 always contain ties)."""
    dev = _dev()
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.receptive_field import GraphBatch
    flat = G.load("g3_molkgnnnet.npz")
    model, state = _net_from_golden(flat, dev)
    model.gnn.set_variant(variant)
    b = GraphBatch(**{k[len("in_"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("in_")})
    b.num_graphs = 3
    bd = b.to(dev)
    emb = model(bd)
    (emb * torch.from_numpy(flat["cotangent"]).to(dev)).sum().backward()
    # replay the layers to collect the build's permutation choices, and check them against the oracle layer by layer
    plan = plan_from_data(bd)
    ostate = {k: v.clone() for k, v in state.items()}
    forced = []
    with torch.no_grad():
        h = model.node_batch_norm(bd.x)
        h_o = O.batch_norm(b.x, ostate["node_batch_norm.weight"], ostate["node_batch_norm.bias"],
                           ostate["node_batch_norm.running_mean"], ostate["node_batch_norm.running_var"], False)
        for i, layer in enumerate(model.gnn.layers):
            params, E = layer._bank_params("train", h)
            sim, saved = Fn.kernelsetconv_details(h, plan, i == 2, params, E, variant)
            idx = [None if s[0] is None else s[0].cpu().long() for s in saved]
            forced.append(idx)
            per_degree = O.kernelset_params(ostate, f"gnn.layers.{i}.")
            assert O.kernelset_tie_aware_mismatch(per_degree, h_o, b, i == 2, sim.cpu(), idx) == 0, f"layer {i}"
            sim_o = O.kernelsetconv(per_degree, h_o, b, i == 2, form="faithful", forced_idx=idx)
            assert torch.allclose(sim.cpu(), sim_o, atol=FWD_TOL, rtol=0)
            h = Fn.propagate_add(sim, plan, out_pad=(-sim.shape[1]) % 4)
            h_o = O.propagate_add(b.edge_index, sim_o)
            assert torch.allclose(h.cpu(), h_o, atol=2e-5, rtol=0)
    ostate = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in ostate.items()}
    emb_o = O.molkgnnnet(ostate, b, 3, training_bn=False, form="faithful", forced_idx=forced)
    assert torch.allclose(emb.detach().cpu(), emb_o, atol=5e-5, rtol=1e-5)
    (emb_o * torch.from_numpy(flat["cotangent"])).sum().backward()
    checked = 0
    for nm, prm in model.named_parameters():
        ref = ostate[nm].grad
        if prm.grad is None:
            assert ref is None or float(ref.abs().max()) == 0.0, nm
            continue
        assert torch.allclose(prm.grad.cpu(), ref, atol=5e-5, rtol=1e-3), (nm, float((prm.grad.cpu() - ref).abs().max()))
        checked += 1
    assert checked > 40


@pytest.mark.parametrize("counts,layers,hidden,mols", [((5, 10, 15, 25), 3, 32, 40), ((16, 32, 48, 64), 3, 64, 40), ((1, 1, 1, 1), 4, 32, 40),
                                                       ((10, 20, 30, 50), 3, 32, 1), ((10, 20, 30, 50), 3, 32, 2)])
def test_three_layer_network_other_reference_configurations(counts, layers, hidden, mols):
    """The reference takes any ``--num_kernel{1..4}_{1hop,Nhop}`` / ``--num_layers`` / ``--hidden_dim`` (MolKGNNNet.py:162-174;
    its sweep launcher runs (1, 1, 1, 1) x 4 layers, utils/scheduler-barium-kgnn.py:181-185).  The whole network -- batch
    norm, every layer on the streamed kernels (55-, 160-, 4-wide N-hop rows), propagate, readout -- against the oracle
    evaluated with the build's own permutation choices, layer by layer and end to end, forward and every parameter gradient.
    ``mols`` = 1 / 2: the smallest batches there are (a single molecule: one or two 16-atom tiles per degree, a degree may be
    absent altogether -- its banks then receive no gradient, as in the reference's autograd)."""
    dev = _dev()
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    torch.manual_seed(sum(counts) + layers)
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    model = MolKGNNNet(num_layers=layers, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=hidden,
                       **dict(zip(names, counts * 2)))
    with torch.no_grad():                                    # non-trivial running statistics for the eval-mode batch norm
        model.node_batch_norm.running_mean.normal_(0.0, 0.3)
        model.node_batch_norm.running_var.uniform_(0.5, 1.5)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    b = make_batch(mols, seed=sum(counts) + mols, duplicate_fraction=0.1)
    b.num_graphs = mols
    bd = b.to(dev)
    cot = torch.randn(mols, hidden, generator=torch.Generator().manual_seed(1))
    from molkgnn_amd import readout as R0
    took = []
    orig_blocks, orig_dense = R0._ReadoutBlocksFn.apply, R0._ReadoutFn.apply
    R0._ReadoutBlocksFn.apply = staticmethod(lambda *a: (took.append("blocks"), orig_blocks(*a))[1])
    R0._ReadoutFn.apply = staticmethod(lambda *a: (took.append("dense"), orig_dense(*a))[1])
    try:
        emb = model(bd)
    finally:
        R0._ReadoutBlocksFn.apply, R0._ReadoutFn.apply = orig_blocks, orig_dense
    # a HIP readout for every one of these shapes: the dense kernels up to 128 columns, the block-row form (K <= 255) beyond
    assert took == (["blocks"] if sum(counts) > 128 else ["dense"]), took
    (emb * cot.to(dev)).sum().backward()
    plan = plan_from_data(bd)
    ostate = {k: v.clone() for k, v in state.items()}
    forced = []
    lib_launches = []
    from molkgnn_amd import readout as R
    with torch.no_grad():
        # (the model's own batch norm operator, not torch's: with duplicated neighbour rows an ulp in x decides
        # mathematically tied orders the other way, and the choices replayed here must be the model's)
        h = R.batch_norm(bd.x, model.node_batch_norm, None)
        h_o = O.batch_norm(b.x, ostate["node_batch_norm.weight"], ostate["node_batch_norm.bias"],
                           ostate["node_batch_norm.running_mean"], ostate["node_batch_norm.running_var"], False)
        for i, layer in enumerate(model.gnn.layers):
            params, E = layer._bank_params("train", h)
            sim, saved = Fn.kernelsetconv_details(h, plan, i == layers - 1, params, E, "mfma")     # (fails if a degree is not covered)
            idx = [None if s[0] is None else s[0].cpu().long() for s in saved]
            forced.append(idx)
            per_degree = O.kernelset_params(ostate, f"gnn.layers.{i}.")
            assert O.kernelset_tie_aware_mismatch(per_degree, h_o, b, i == layers - 1, sim.cpu(), idx) == 0, f"layer {i}"
            sim_o = O.kernelsetconv(per_degree, h_o, b, i == layers - 1, form="faithful", forced_idx=idx)
            assert torch.allclose(sim.cpu(), sim_o, atol=FWD_TOL, rtol=0)
            h = Fn.propagate_add(sim, plan, out_pad=(-sim.shape[1]) % 4)
            h_o = O.propagate_add(b.edge_index, sim_o)
    ostate = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in ostate.items()}
    emb_o = O.molkgnnnet(ostate, b, layers, training_bn=False, form="faithful", forced_idx=forced)
    scale = max(1.0, float(emb_o.detach().abs().max()))
    assert float((emb.detach().cpu() - emb_o.detach()).abs().max()) <= 5e-5 * scale
    (emb_o * cot).sum().backward()
    checked = 0
    for nm, prm in model.named_parameters():
        ref = ostate[nm].grad
        if prm.grad is None:
            assert ref is None or float(ref.abs().max()) == 0.0, nm
            continue
        assert float((prm.grad.cpu() - ref).abs().max()) <= 5e-5 * max(1.0, float(ref.abs().max())) + 1e-3 * float(ref.abs().max()), \
            (nm, float((prm.grad.cpu() - ref).abs().max()), float(ref.abs().max()))
        checked += 1
    present = sum(1 for d in range(1, 5) if getattr(b, f"selected_index_deg{d}").numel() > 0)
    assert checked >= 6 * present * layers


def test_fullsize_seeded_model_matches_reference_output():
    """Same seed -> same 130 090 parameters as the reference (init order)
    and, on a tie-free first layer, the same first-layer scores."""
    dev = _dev()
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    from molkgnn_amd.receptive_field import GraphBatch
    flat = G.load("g7_fullsize.npz")
    torch.manual_seed(int(flat["seed"]))
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    model = MolKGNNNet(num_layers=3, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32,
                       **dict(zip(names, (10, 20, 30, 50) * 2)))
    assert sum(p.numel() for p in model.parameters()) == int(flat["num_params"]) == 130090
    assert [n for n, _ in model.named_parameters()] == list(flat["param_names"])
    sums = np.array([float(p.detach().double().sum()) for p in model.parameters()])
    assert np.array_equal(sums, flat["param_sums"])
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    b = GraphBatch(**{k[len("in_"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("in_")})
    b.num_graphs = 4
    bd = b.to(dev)
    emb = model(bd)
    assert emb.shape == flat["graph_embedding"].shape
    # first layer (tie-free on random features): the reference's own scores, 1e-5
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_data
    plan = plan_from_data(bd)
    forced = []
    with torch.no_grad():
        h = model.node_batch_norm(bd.x)
        for i, layer in enumerate(model.gnn.layers):
            params, E = layer._bank_params("train", h)
            sim, saved = Fn.kernelsetconv_details(h, plan, i == 2, params, E)
            forced.append([None if s_[0] is None else s_[0].cpu().long() for s_ in saved])
            if i == 0:
                assert torch.allclose(sim.cpu(), torch.from_numpy(flat["layer0_sim_sc"]), atol=FWD_TOL, rtol=0)
            h = Fn.propagate_add(sim, plan, out_pad=(-sim.shape[1]) % 4)
    # the embedding: the oracle replayed with the build's permutation choices (layers >= 1 contain structural ties) ...
    emb_o = O.molkgnnnet(state, b, 3, training_bn=False, form="faithful", forced_idx=forced)
    assert torch.allclose(emb.detach().cpu(), emb_o, atol=5e-5, rtol=1e-5), float((emb.detach().cpu() - emb_o).abs().max())
    # ... and where the build chose exactly what the oracle's argmax chooses, the reference's own embedding
    h_o = O.batch_norm(b.x, state["node_batch_norm.weight"], state["node_batch_norm.bias"],
                       state["node_batch_norm.running_mean"], state["node_batch_norm.running_var"], False)
    same = True
    for i in range(3):
        got = []
        sim_o = O.kernelsetconv(O.kernelset_params(state, f"gnn.layers.{i}."), h_o, b, i == 2, idx_out=got)
        same = same and all((a is None and c is None) or torch.equal(a, c) for a, c in zip(got, forced[i]))
        h_o = O.propagate_add(b.edge_index, sim_o)
    if same:
        assert torch.allclose(emb.detach().cpu(), torch.from_numpy(flat["graph_embedding"]), atol=5e-5, rtol=1e-5)


def test_error_behaviour():
    dev = _dev()
    from molkgnn_amd.kernels import KernelConv, KernelSetConv
    from molkgnn_amd._lib import MolKGNNLibraryError
    with pytest.raises(Exception, match="not specified"):
        KernelConv(L=3)
    conv = KernelConv(L=3, D=3, num_supports=2, node_attr_dim=5, edge_attr_dim=2).to(dev)
    with pytest.raises(Exception, match="2D, but the kernel is 3D"):
        conv(False, x_focal=torch.randn(4, 5, device=dev), p_focal=torch.randn(4, 2, device=dev),
             x_neighbor=torch.randn(4, 2, 5, device=dev), p_neighbor=torch.randn(4, 2, 2, device=dev),
             edge_attr_neighbor=torch.randn(4, 2, 2, device=dev))
    cpu_conv = KernelConv(L=3, D=3, num_supports=2, node_attr_dim=5, edge_attr_dim=2)
    with pytest.raises(MolKGNNLibraryError, match="no CPU fallback"):
        cpu_conv(False, x_focal=torch.randn(4, 5), p_focal=torch.randn(4, 3), x_neighbor=torch.randn(4, 2, 5),
                 p_neighbor=torch.randn(4, 2, 3), edge_attr_neighbor=torch.randn(4, 2, 2))
    ksc = KernelSetConv(2, 2, 2, 2, D=3, node_attr_dim=5, edge_attr_dim=2)
    assert ksc.get_num_kernel() == 8 and ksc.num_kernel_list == [2, 2, 2, 2]
    assert all(k is None for k in ksc.fixed_kernelconv_set)


def test_unusual_shapes_generic_path():
    """MolGCN's default widths (x_dim=5, edge_attr_dim=1) and an odd kernel count run on the
    generic kernels and agree with the oracle."""
    dev = _dev()
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(5, seed=9)
    torch.manual_seed(5)
    ksc = KernelSetConv(3, 1, 7, 2, D=3, node_attr_dim=5, edge_attr_dim=1)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(b.x.shape[0], 5, generator=g)
    for d in range(1, 5):
        e = getattr(b, f"nei_edge_attr_deg{d}")
        setattr(b, f"nei_edge_attr_deg{d}", e[..., :1].contiguous() + 0.5 if e.numel() else e)
    b.x = x
    state = {k: v.detach().clone() for k, v in ksc.state_dict().items()}
    ref = O.kernelsetconv(O.kernelset_params(state), x, b, True)
    bd = b.to(dev)
    out = ksc.to(dev)(is_last_layer=True, data=bd, save_score=False)
    assert torch.allclose(out.cpu(), ref, atol=FWD_TOL, rtol=0)


def test_propagate_hands_row_norms_to_next_layer():
    """propagate_add emits 1/max(|h_n|, eps) with h; the next convolution uses it only while h is unmodified."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    batch = make_batch(24, seed=5, device="cuda")
    plan = plan_from_data(batch)
    g = torch.Generator().manual_seed(3)
    for width in (110, 37):                       # even width: 8-byte path; odd width: scalar path
        v = torch.randn(batch.x.shape[0], width, generator=g).cuda()
        h = Fn.propagate_add(v, plan, out_pad=(-width) % 4)
        ref = torch.zeros_like(v).index_add_(0, batch.edge_index[1], v[batch.edge_index[0]])
        assert torch.allclose(h, ref, atol=1e-5)
        inv = Fn._handed_inv_norm(h)
        assert inv is not None
        assert torch.allclose(inv, 1.0 / h.norm(dim=1).clamp_min(1e-8), rtol=1e-5)
        assert torch.equal(inv, Fn.row_inv_norm(h))          # same summation order in both kernels
        h.mul_(2.0)
        assert Fn._handed_inv_norm(h) is None


def _torch_readout(h, w1, b1, w2, b2, keep, batch, size):
    z = torch.nn.functional.linear(h, w1, b1)
    z = z * torch.sigmoid(z)
    if keep is not None:
        z = z * keep
    z = torch.nn.functional.linear(z, w2, b2)
    return torch.zeros(size, w2.shape[0], device=h.device).index_add_(0, batch, z)


@pytest.mark.parametrize("F,H,G,bias,drop", [(110, 32, 32, True, False), (110, 32, 32, True, True), (28, 5, 7, False, False),
                                              (70, 64, 3, True, True), (128, 33, 64, True, False)])
def test_readout_matches_torch_formula(F, H, G, bias, drop):
    """pool(lin2(dropout(swish(lin1(h))))) (reference MolKGNNNet.py:144-146) against the same formula in
    PyTorch fp32 operators: forward 1e-5 relative to the output scale, all five gradients."""
    from molkgnn_amd import readout as R
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    b = make_batch(37, seed=F + H, device=dev, with_receptive_fields=False)
    n, size = b.x.shape[0], 37
    g = torch.Generator().manual_seed(F * 1000 + H)
    rnd = lambda *s: torch.randn(*s, generator=g)
    h = rnd(n, F).to(dev).requires_grad_(True)
    w1 = (rnd(H, F) * F ** -0.5).to(dev).requires_grad_(True)
    w2 = (rnd(G, H) * H ** -0.5).to(dev).requires_grad_(True)
    b1 = rnd(H).to(dev).requires_grad_(True) if bias else None
    b2 = rnd(G).to(dev).requires_grad_(True) if bias else None
    keep = ((torch.rand(n, H, generator=g) > 0.25).float() / 0.75).to(dev) if drop else None
    cot = rnd(size, G).to(dev)
    seg = R.molecule_segments(b.batch, size)
    assert seg.sorted
    out = R._ReadoutFn.apply(h, w1, b1, w2, b2, keep, seg)
    leaves = [t for t in (h, w1, b1, w2, b2) if t is not None]
    got = torch.autograd.grad((out * cot).sum(), leaves)
    ref = _torch_readout(h, w1, b1, w2, b2, keep, b.batch, size)
    want = torch.autograd.grad((ref * cot).sum(), leaves)
    scale = float(ref.detach().abs().max())
    assert float((out - ref).detach().abs().max()) <= 1e-5 * max(scale, 1.0)
    for a, w in zip(got, want):
        assert float((a - w).abs().max()) <= 2e-5 * max(float(w.abs().max()), 1.0), (a.shape, float((a - w).abs().max()))


@pytest.mark.parametrize("counts,H,G,bias,drop,mols", [((10, 20, 30, 50), 32, 32, True, True, 300), ((5, 10, 15, 25), 33, 7, False, False, 37),
                                                       ((1, 1, 1, 1), 5, 64, True, False, 60), ((16, 32, 48, 64), 64, 32, True, True, 90)])
def test_block_row_readout_matches_torch_formula(counts, H, G, bias, drop, mols):
    """readout.readout_blocks (mkgnn_readout_blocks_*: lin1 on every atom's own column block first, the propagate step on
    H-wide rows) against  pool(lin2(dropout(swish(lin1(propagate(sim))))))  in PyTorch fp32 operators (KernelLayer.py:
    119-123, MolKGNNNet.py:144-146): forward 1e-5 of the output scale, the gradient of sim on every atom's own block and all
    four parameter gradients 2e-5 relative.  NaN is written everywhere outside the blocks: nothing may read it."""
    from molkgnn_amd import readout as R
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    b = make_batch(mols, seed=sum(counts) + H, device=dev)
    plan = plan_from_data(b)
    n, K = b.x.shape[0], sum(counts)
    g = torch.Generator().manual_seed(K * 100 + H)
    rnd = lambda *s: torch.randn(*s, generator=g)
    K4 = K + (-K) % 4
    # block rows: atom n holds values only in the columns of its degree
    deg = torch.zeros(n, dtype=torch.long)
    for d in range(1, 5):
        deg[getattr(b, f"selected_index_deg{d}").cpu()] = d
    offs = [0, counts[0], counts[0] + counts[1], counts[0] + counts[1] + counts[2], K]
    mask = torch.zeros(n, K, dtype=torch.bool)
    for d in range(1, 5):
        mask[deg == d, offs[d - 1]:offs[d]] = True
    dense = torch.where(mask, rnd(n, K), torch.zeros(()))
    store = torch.full((n, K4), float("nan"))
    store[:, :K] = torch.where(mask, dense, torch.full((), float("nan")))
    sim = store.to(dev)[:, :K].requires_grad_(True)
    w1 = (rnd(H, K) * K ** -0.5).to(dev).requires_grad_(True)
    w2 = (rnd(G, H) * H ** -0.5).to(dev).requires_grad_(True)
    b1 = rnd(H).to(dev).requires_grad_(True) if bias else None
    b2 = rnd(G).to(dev).requires_grad_(True) if bias else None
    keep = ((torch.rand(n, H, generator=g) > 0.25).float() / 0.75).to(dev) if drop else None
    cot = rnd(mols, G).to(dev)
    seg = R.molecule_segments(b.batch, mols)
    assert seg.sorted and R.readout_blocks_supported(K, H, G, counts)
    out = R._ReadoutBlocksFn.apply(sim, w1, b1, w2, b2, keep, seg, plan, tuple(counts))
    with torch.no_grad():                    # inference: no gate is left in place of pre, the same output bit for bit
        out_ng = R._ReadoutBlocksFn.apply(sim, w1, b1, w2, b2, keep, seg, plan, tuple(counts))
    assert torch.equal(out_ng, out.detach())
    leaves = [t for t in (sim, w1, b1, w2, b2) if t is not None]
    got = torch.autograd.grad((out * cot).sum(), leaves)
    # the reference: dense sim (zeros outside the blocks) -> propagate -> readout, in PyTorch operators
    sim_ref = dense.to(dev).requires_grad_(True)
    h = torch.zeros(n, K, device=dev).index_add(0, b.edge_index[1], sim_ref[b.edge_index[0]])
    ref = _torch_readout(h, w1, b1, w2, b2, keep, b.batch, mols)
    want = torch.autograd.grad((ref * cot).sum(), [sim_ref] + leaves[1:])
    scale = float(ref.detach().abs().max())
    assert float((out - ref).detach().abs().max()) <= 1e-5 * max(scale, 1.0)
    m = mask.to(dev)
    gs, ws_ = got[0], want[0]
    assert float((gs[m] - ws_[m]).abs().max()) <= 2e-5 * max(float(ws_.abs().max()), 1.0)
    for a, w in zip(got[1:], want[1:]):
        assert float((a - w).abs().max()) <= 2e-5 * max(float(w.abs().max()), 1.0), (a.shape, float((a - w).abs().max()))


def test_model_with_block_row_readout_matches_the_propagate_then_readout_model(monkeypatch):
    """MolKGNNNet with the last propagate left to the readout (MKGNN_PROJECT_FIRST) against the same model with
    propagate -> readout: the same embedding to 1e-5 and every parameter's gradient to 2e-5 of its scale (the sums are
    re-associated, nothing else), in training mode with batch-norm statistics, through the fused convolution backward."""
    import copy
    from molkgnn_amd import MolKGNNNet as M
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, backward
    dev = _dev()
    torch.manual_seed(4)
    b = make_batch(700, seed=14).to(dev)
    b.y = (torch.arange(700, device=dev) % 4 == 0).long()
    m0 = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev)
    m1 = copy.deepcopy(m0)
    res = []
    for model, flag in ((m0, '0'), (m1, '1')):
        monkeypatch.setattr(M, "_PROJECT_FIRST", flag)
        model.zero_grad(set_to_none=True)
        emb = model.gnn_model(b)
        loss = model.loss(b)
        backward(loss)
        torch.cuda.synchronize()
        res.append((emb.detach().clone(), float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    (e0, l0, g0), (e1, l1, g1) = res
    assert float((e0 - e1).abs().max()) <= 1e-5 * max(1.0, float(e0.abs().max()))
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0))
    assert g0.keys() == g1.keys()
    for n, gw in g0.items():
        assert float((g1[n] - gw).abs().max()) <= 2e-5 * max(float(gw.abs().max()), 1e-3) + 1e-7, (n, float((g1[n] - gw).abs().max()))


def test_readout_module_paths():
    """The module-level entry point: fused path, dropout in eval mode, an unsorted batch vector (PyTorch
    operators on the GPU) and an empty molecule in the middle of the batch."""
    from molkgnn_amd import readout as R
    dev = _dev()
    torch.manual_seed(0)
    lin1, lin2, drop = torch.nn.Linear(110, 32).to(dev), torch.nn.Linear(32, 32).to(dev), torch.nn.Dropout(0.25).eval()
    batch = torch.tensor([0] * 5 + [1] * 20 + [3] * 7, device=dev)          # molecule 2 has no atoms
    h = torch.randn(32, 110, device=dev)
    ref = _torch_readout(h, lin1.weight, lin1.bias, lin2.weight, lin2.bias, None, batch, 4)
    out = R.readout(h, lin1, lin2, drop, batch, 4)
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)
    assert float(out[2].detach().abs().max()) == 0.0
    perm = torch.randperm(32, device=dev)
    out_p = R.readout(h[perm], lin1, lin2, drop, batch[perm].contiguous(), 4)
    assert torch.allclose(out_p, ref, atol=1e-5, rtol=1e-5)
    drop.train()
    torch.manual_seed(1)
    o1 = R.readout(h, lin1, lin2, drop, batch, 4)
    assert not torch.allclose(o1, ref, atol=1e-3)                            # multipliers were applied


@pytest.mark.parametrize("C,n,affine", [(28, 1000, True), (28, 37, True), (7, 513, True), (130, 300, False)])
def test_batch_norm_matches_torch(C, n, affine):
    """node_batch_norm (reference MolKGNNNet.py:115): training and eval forward, running statistics and all
    three gradients against torch.nn.BatchNorm1d."""
    from molkgnn_amd import readout as R
    dev = _dev()
    torch.manual_seed(C)
    x = (torch.randn(n, C, device=dev) * 3 + 1.5)
    cot = torch.randn(n, C, device=dev)
    mine, ref = torch.nn.BatchNorm1d(C, affine=affine).to(dev), torch.nn.BatchNorm1d(C, affine=affine).to(dev)
    if affine:
        with torch.no_grad():
            mine.weight.uniform_(0.5, 1.5); mine.bias.normal_()
            ref.weight.copy_(mine.weight); ref.bias.copy_(mine.bias)
    for mode in ("train", "train", "eval"):
        mine.train(mode == "train"); ref.train(mode == "train")
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = R.batch_norm(xa, mine), ref(xb)
        assert torch.allclose(ya, yb, atol=2e-5, rtol=1e-5)
        (ya * cot).sum().backward(); (yb * cot).sum().backward()
        assert torch.allclose(xa.grad, xb.grad, atol=2e-5, rtol=1e-4)
        if affine:
            assert torch.allclose(mine.weight.grad, ref.weight.grad, atol=1e-3, rtol=1e-4)
            assert torch.allclose(mine.bias.grad, ref.bias.grad, atol=1e-3, rtol=1e-4)
            mine.zero_grad(); ref.zero_grad()
        assert torch.allclose(mine.running_mean, ref.running_mean, atol=1e-6, rtol=1e-5)
        assert torch.allclose(mine.running_var, ref.running_var, atol=1e-5, rtol=1e-5)
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("banks", [(10, 20, 30, 50), (10, 20, 30, 100), (60, 20, 30, 50)])
def test_backward_under_graph_capture_matches_eager(banks):
    """Inside a hipGraph capture the backward call runs its x-gradient and bank-gradient chains on two streams
    (kgnn_capi.hip ForkJoin); the replayed graph must reproduce the eager gradients bit for bit."""
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(11)
    b = make_batch(200, seed=21, device=dev)
    plan = plan_from_data(b)
    _ = plan.scatter
    # (a 100- or 60-kernel bank is outside the MFMA / LDS backward kernels: that degree's gradients then come from the
    # generic kernels on the captured stream while the other degrees use the two-stream split)
    layer = KernelSetConv(*banks, D=3, node_attr_dim=28, edge_attr_dim=7).to(dev)
    x = b.x.clone().requires_grad_(True)
    cot = torch.randn(b.x.shape[0], sum(banks), device=dev)

    def step():
        for p in layer.parameters():
            p.grad = None
        x.grad = None
        out = layer._run(x, plan, False)
        (out * cot).sum().backward()
        return out

    out_e = step().detach().clone()
    eager = [x.grad.clone()] + [p.grad.clone() for p in layer.parameters() if p.grad is not None]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = step()
    x.grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    captured = [x.grad] + [p.grad for p in layer.parameters() if p.grad is not None]
    assert torch.equal(out_g, out_e)
    assert len(captured) == len(eager)
    for g, e in zip(captured, eager):
        assert torch.equal(g, e)


def test_csr_passes_long_and_empty_segments():
    """The pipelined CSR kernels fetch four rows per segment in the steady state and loop for longer ones: a hub
    atom with nine bonds (not in any degree bucket, but a neighbour of nine degree-1 atoms and the target of nine
    edges), isolated atoms and widths that select every lanes-per-row variant."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_lists
    from molkgnn_amd.receptive_field import build_receptive_fields
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    # atoms 0..8 leaves of hub 9; 10-11 isolated; 12-13-14 a chain; 15..23 a second star around 24 plus a tail 24-25
    pairs = [(i, 9) for i in range(9)] + [(12, 13), (13, 14)] + [(i, 24) for i in range(15, 24)] + [(24, 25)]
    ei = torch.tensor([[a, b] for a, b in pairs for (a, b) in ((a, b), (b, a))]).t().contiguous()
    n = 26
    for width in (110, 28, 60, 200, 7):
        w4 = width + (-width) % 4
        store = torch.zeros(n, w4)
        store[:, :width] = torch.randn(n, width, generator=g)
        v = store.to(dev)[:, :width]
        empty, emptyf = torch.zeros(0, dtype=torch.long, device=dev), torch.zeros(0, device=dev)
        plan = plan_from_lists(n, [emptyf] * 4, [emptyf] * 4, [emptyf] * 4, [empty] * 4, [empty] * 4, ei.to(dev))
        h = Fn.propagate_add(v, plan, out_pad=(-width) % 4)
        ref = torch.zeros(n, width).index_add_(0, ei[1], store[:, :width][ei[0]])
        assert torch.allclose(h.cpu(), ref, atol=1e-5), width
        assert float(h[10].abs().max()) == 0.0 and float(h[11].abs().max()) == 0.0
        inv = Fn._handed_inv_norm(h)
        assert torch.allclose(inv.cpu(), 1.0 / ref.norm(dim=1).clamp_min(1e-8), rtol=1e-5), width
    # backward gather: the hubs are neighbours in nine (degree-1 focal) roles each -> scatter segments of length 9
    x = torch.randn(n, 28, generator=g)
    p = torch.randn(n, 3, generator=g)
    ea = torch.rand(ei.shape[1], 7, generator=g)
    ea[1::2] = ea[0::2]
    from molkgnn_amd.receptive_field import GraphBatch
    fields = build_receptive_fields(x, p, ei, ea)
    b = GraphBatch(x=x, p=p, edge_index=ei, edge_attr=ea, batch=torch.zeros(n, dtype=torch.long), **fields).to(dev)
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    torch.manual_seed(3)
    layer = KernelSetConv(4, 3, 2, 2, D=3, node_attr_dim=28, edge_attr_dim=7).to(dev)
    plan = plan_from_data(b)
    grads = {}
    for variant in VARIANTS:
        layer.variant = variant
        xg = b.x.clone().requires_grad_(True)
        out = layer._run(xg, plan, False)
        (out * torch.arange(out.numel(), device=dev).reshape(out.shape).float().cos()).sum().backward()
        grads[variant] = xg.grad.clone()
    assert torch.allclose(grads["generic"], grads["mfma"], atol=2e-5, rtol=1e-4)
    assert float(grads["mfma"][9].abs().max()) > 0.0          # the hub received its nine contributions


@pytest.mark.parametrize("B,H", [(4096, 32), (37, 5), (1, 40)])
def test_bce_head_loss_matches_torch(B, H):
    """ffn + BCEWithLogitsLoss (reference model.py:147-148, 190-198) against the PyTorch operators."""
    from molkgnn_amd.readout import bce_head_loss
    dev = _dev()
    torch.manual_seed(B + H)
    ffn = torch.nn.Linear(H, 1).to(dev)
    emb = (torch.randn(B, H, device=dev) * 2).requires_grad_(True)
    y = (torch.rand(B, device=dev) < 0.3).long()
    loss = bce_head_loss(emb, ffn, y)
    got = torch.autograd.grad(loss * 1.7, [emb, ffn.weight, ffn.bias])
    ref = torch.nn.BCEWithLogitsLoss()(ffn(emb).view(-1), y.view(-1).float())
    want = torch.autograd.grad(ref * 1.7, [emb, ffn.weight, ffn.bias])
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * max(1.0, abs(float(ref.detach())))
    for a, w in zip(got, want):
        assert a.shape == w.shape
        assert torch.allclose(a, w, atol=1e-7, rtol=2e-5), float((a - w).abs().max())


@pytest.mark.parametrize("B,H,p", [(4096, 32, 0.25), (37, 5, 0.0), (300, 40, 0.5)])
def test_fused_head_gives_the_separate_backward_bit_for_bit(B, H, p, monkeypatch):
    """mkgnn_bce_head_fused (forward + the gradients for d loss = 1 in the same two launches): the same loss (to the last
    bit or two) and the same gradients, bit for bit, as the separate forward and backward launches -- seeded through train.backward (the registered
    ones tensor: the backward launches nothing) and through an arbitrary upstream gradient (scaled)."""
    from molkgnn_amd import readout as R
    from molkgnn_amd.train import backward
    dev = _dev()
    torch.manual_seed(B + H)
    ffn = torch.nn.Linear(H, 1).to(dev)
    emb0 = torch.randn(B + 3, H, device=dev) * 2
    y = (torch.rand(B, device=dev) < 0.3).long()

    def run(split, scale):
        monkeypatch.setattr(R, "_SPLIT_HEAD", split)
        R.reset_head_rng(dev, seed=1234)
        emb = emb0.clone().requires_grad_(True)
        ffn.zero_grad(set_to_none=True)
        loss = R.bce_head_loss(emb, ffn, y, dropout_p=p, n_rows=B)
        if scale is None:
            backward(loss)
        else:
            (loss * scale).backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), emb.grad.clone(), ffn.weight.grad.clone(), ffn.bias.grad.clone()

    for scale in (None, 1.7):
        a, c = run(True, scale), run(False, scale)
        assert abs(float(a[0]) - float(c[0])) <= 1e-6 * abs(float(c[0]))      # (the two final kernels sum the block partials in different fixed orders)
        if scale is None:
            assert all(torch.equal(u, v) for u, v in zip(a[1:], c[1:]))
        else:                                   # (g * 1.7 after the sum against 1.7 folded into every term)
            assert all(float((u - v).abs().max()) <= 2e-6 * float(v.abs().max()) for u, v in zip(a[1:], c[1:]))
        assert float(a[1][B:].abs().max()) == 0.0 and float(c[1][B:].abs().max()) == 0.0


def _oracle_check_both_variants(layer, state, cpu_batch, plan, x_cpu, store, width, last, cot_cpu, dev, what):
    """Forward (two-part tie criterion) and every gradient of the generic AND of the fast kernels against the oracle
    replayed with that variant's own permutation choices: no HIP-vs-HIP comparison, no skipped gradient checks."""
    from tests.test_scale_parity import _run_build, _close
    per_degree = O.kernelset_params(state)
    n = x_cpu.shape[0]
    for variant, bwd in (("generic", "generic"), ("mfma", "auto")):
        out, idx, gx, grads = _run_build(layer, store, width, plan, last, variant, bwd, cot_cpu.to(dev))
        assert torch.isfinite(out).all() and torch.isfinite(gx).all(), (what, variant)
        bad, _ = O.kernelset_forced_mismatch(per_degree, x_cpu, cpu_batch, last, out, idx, form="cosmat", tol=FWD_TOL)
        assert bad == 0, (what, variant, bad)
        _, gx_o, grads_o = O.kernelset_gradients(state, x_cpu, cpu_batch, last, cot_cpu, forced_idx=idx, form="cosmat")
        _close(gx, gx_o, (what, variant, "grad_x"), 3e-5, 1e-3)
        scale = max(1.0, (n / 64.0) ** 0.5)
        for name, ref in grads_o.items():
            if ref is None or ref.numel() == 0:
                continue
            _close(grads[name].reshape(ref.shape), ref, (what, variant, name), 2e-5 * scale, 1e-3)


@pytest.mark.parametrize("width,last", [(55, False), (12, False), (96, True), (40, False), (33, False), (110, True)])
def test_large_batch_other_widths_against_the_oracle(width, last):
    """Tile boundaries, multi-tile waves, partial last tiles and the two-stream / packed-row code paths only show up at
    scale: ~13 k atoms, row widths that take the every-chunk-masked forward and (odd widths) mixed fast / generic
    backward kernels.  Each variant is compared with the ORACLE evaluated with that variant's own permutation choices
    (tests/test_scale_parity.py does the same for the benchmark's widths 28 and 110 with the fast backward forced)."""
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(width + int(last))
    cpu = make_batch(517, seed=77 + width)
    bd = cpu.to(dev)
    plan = plan_from_data(bd)
    layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7)
    state = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    n = cpu.x.shape[0]
    g = torch.Generator().manual_seed(width)
    x_cpu = torch.randn(n, width, generator=g)
    cot_cpu = torch.randn(n, 110, generator=g)
    store = torch.zeros(n, width + (-width) % 4, device=dev)
    store[:, :width] = x_cpu.to(dev)
    _oracle_check_both_variants(layer, state, cpu, plan, x_cpu, store, width, last, cot_cpu, dev, (width, last))


def test_receptive_field_builder_hip_matches_torch_builder():
    """mkgnn_rf_count / mkgnn_rf_fill (reference wrapper.py:559-672) against the torch builder, which the CPU suite
    checks against a per-atom brute force: exact equality of all 20 tensors, on a synthetic batch, on the same batch
    with the bonds shuffled (sources no longer contiguous in the edge list), and on a graph with hub atoms
    (out-degree 9, in no bucket), isolated atoms and an absent degree."""
    from molkgnn_amd.receptive_field import build_receptive_fields, build_receptive_fields_hip
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    b = make_batch(300, seed=4, device=dev, with_receptive_fields=False)
    cases = [(b.x, b.p, b.edge_index, b.edge_attr)]
    g = torch.Generator().manual_seed(1)
    nb = b.edge_index.shape[1] // 2
    perm = torch.randperm(nb, generator=g).to(dev)
    eperm = torch.stack([2 * perm, 2 * perm + 1], dim=1).reshape(-1)
    cases.append((b.x, b.p, b.edge_index[:, eperm].contiguous(), b.edge_attr[eperm].contiguous()))
    pairs = [(i, 9) for i in range(9)] + [(12, 13), (13, 14)] + [(i, 24) for i in range(15, 24)] + [(24, 25)]
    ei = torch.tensor([[a, c] for a, c in pairs for (a, c) in ((a, c), (c, a))]).t().contiguous().to(dev)
    ea = torch.rand(ei.shape[1] // 2, 7, generator=g).repeat_interleave(2, dim=0).to(dev)
    cases.append((torch.randn(26, 28, device=dev), torch.randn(26, 3, device=dev), ei, ea))
    for x, p, edge_index, edge_attr in cases:
        want = build_receptive_fields(x, p, edge_index, edge_attr)
        got = build_receptive_fields_hip(x, p, edge_index, edge_attr)
        assert set(want) <= set(got)
        for k in want:
            assert want[k].shape == got[k].shape and want[k].dtype == got[k].dtype, k
            assert torch.equal(want[k], got[k]), k
        # the unit-normalised bond rows that ride along: bit for bit what mkgnn_unit_rows8 makes of the raw rows
        from molkgnn_amd.plan import Bucket
        for d in range(1, 5):
            if want[f"selected_index_deg{d}"].numel():
                bk = Bucket(d, got[f"selected_index_deg{d}"], got[f"nei_index_deg{d}"], got[f"nei_edge_attr_deg{d}"],
                            got[f"p_focal_deg{d}"], got[f"nei_p_deg{d}"])
                assert torch.equal(got[f"nei_edge_unit_deg{d}"], bk.e_unit(7)), d


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("MKGNN_FUZZ", "10")))))
def test_random_shapes_against_the_oracle(seed):
    """Random row widths, bond widths, bank sizes (including one kernel, 17 = two column tiles, 50) and batch sizes,
    first / last layer: the fast path (forced in the forward: an unsupported shape would raise) and the generic kernels,
    each against the oracle with its own permutation choices."""
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.receptive_field import GraphBatch, build_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    rng = np.random.default_rng(1000 + seed)
    F = int(rng.choice([2, 6, 16, 18, 28, 30, 44, 64, 80, 110, 112]))
    E = int(rng.integers(1, 9))
    Ls = [int(rng.choice([1, 3, 10, 16, 17, 20, 30, 50])) for _ in range(4)]
    nmol = int(rng.choice([1, 3, 20, 90]))
    last = bool(rng.integers(0, 2))
    topo = make_batch(nmol, seed=seed, with_receptive_fields=False)
    g = torch.Generator().manual_seed(seed)
    n, m = topo.x.shape[0], topo.edge_index.shape[1]
    x = torch.randn(n, F, generator=g)
    ea = torch.rand(m // 2, E, generator=g).repeat_interleave(2, dim=0)
    fields = build_receptive_fields(x, topo.p, topo.edge_index, ea)
    cpu = GraphBatch(x=x, p=topo.p, edge_index=topo.edge_index, edge_attr=ea, batch=topo.batch, **fields)
    bd = cpu.to(dev)
    torch.manual_seed(seed)
    layer = KernelSetConv(*Ls, D=3, node_attr_dim=F, edge_attr_dim=E)
    state = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    plan = plan_from_data(bd)
    cot = torch.randn(n, sum(Ls), generator=g)
    store = torch.zeros(n, F + (-F) % 4, device=dev)
    store[:, :F] = x.to(dev)
    _oracle_check_both_variants(layer, state, cpu, plan, x, store, F, last, cot, dev, (F, E, Ls, nmol, last))


@pytest.mark.parametrize("nmol,F,last", [(1, 110, False), (2, 28, False), (7, 110, True), (33, 110, False), (33, 28, False),
                                         (150, 110, True)])
def test_streamed_kernels_on_small_and_ragged_batches(nmol, F, last):
    """The reference's own shapes (kernels 10 / 20 / 30 / 50, F = 28 / 110, E = 7: the streamed forward, rows and bank
    kernels) on batches with fewer atom tiles than streams, buckets of one or two atoms and ragged last tiles: both
    variants against the oracle, forward and every gradient."""
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.receptive_field import GraphBatch, build_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    topo = make_batch(nmol, seed=300 + nmol, with_receptive_fields=False)
    g = torch.Generator().manual_seed(nmol)
    n = topo.x.shape[0]
    x = torch.randn(n, F, generator=g)
    fields = build_receptive_fields(x, topo.p, topo.edge_index, topo.edge_attr)
    cpu = GraphBatch(x=x, p=topo.p, edge_index=topo.edge_index, edge_attr=topo.edge_attr, batch=topo.batch, **fields)
    bd = cpu.to(dev)
    torch.manual_seed(nmol)
    layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=F, edge_attr_dim=7)
    state = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev)
    plan = plan_from_data(bd)
    cot = torch.randn(n, 110, generator=g)
    store = torch.zeros(n, F + (-F) % 4, device=dev)
    store[:, :F] = x.to(dev)
    _oracle_check_both_variants(layer, state, cpu, plan, x, store, F, last, cot, dev, (F, nmol, last))


@pytest.mark.parametrize("width,last", [(28, False), (110, False), (110, True), (55, False)])
def test_bf16_similarity_variant_tracks_fp32(width, last):
    """variant="bf16" (BASELINE configs[4], SURVEY 8c config 5: bf16 similarity path, fp32 accumulate, parity relaxed to
    bf16 tolerance, argmax may legitimately differ).  Where both paths choose the same neighbour order the scores agree
    to 1e-2 (3e-3 on average); a different order is chosen for few (atom, kernel) pairs -- near-ties -- and there the
    support score still agrees (it is a maximum over orders), only the bond score follows the other order.  The
    backward (fp32, through the saved order) runs and is finite."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(width)
    b = make_batch(200, seed=3 + width, device=dev)
    plan = plan_from_data(b)
    layer = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=width, edge_attr_dim=7).to(dev)
    n = b.x.shape[0]
    store = torch.zeros(n, width + (-width) % 4, device=dev)
    store[:, :width] = torch.randn(n, width, device=dev)
    params, E = layer._bank_params("train", store[:, :width])
    det = {v: Fn.kernelsetconv_details(store[:, :width], plan, last, params, E, v) for v in ("mfma", "bf16")}
    same_total = pairs_total = 0
    for d in range(4):
        bi32, sc32, _ = det["mfma"][1][d]
        bi16, sc16, _ = det["bf16"][1][d]
        if bi32 is None:
            continue
        same = bi32 == bi16                              # [L_d, N_d]
        same_total += int(same.sum()); pairs_total += same.numel()
        assert float((sc32[0] - sc16[0]).abs().max()) <= 1e-2          # support score: a maximum, robust to the choice
        assert float((sc32[1] - sc16[1]).abs().max()) <= 1e-2          # centre score: no choice involved
        assert float((sc32[2] - sc16[2]).abs()[same].max()) <= 1e-6    # bond score: fp32 in both, equal for equal orders
    assert same_total / pairs_total > 0.97, same_total / pairs_total
    diff = (det["mfma"][0] - det["bf16"][0]).abs()
    assert float(diff.mean()) <= 3e-3 and float(diff.max()) > 0.0
    layer.variant = "bf16"
    x = store[:, :width].detach().requires_grad_(True)
    o = layer._run(x, plan, last)
    o.square().sum().backward()
    assert torch.isfinite(o).all() and torch.isfinite(x.grad).all()
    assert all(torch.isfinite(p_.grad).all() for p_ in layer.parameters() if p_.grad is not None)


def test_whole_training_step_under_graph_capture_matches_eager():
    """The complete step (batch norm, three convolution layers, propagate, readout, head + loss, backward, AdamW) captured
    into one hipGraph and replayed -- what bench.py measures -- must leave the same parameters as the eager step: every
    operator is capturable (no host synchronisation, no allocation-dependent state) and deterministic."""
    import copy
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, configure_optimizer
    dev = _dev()
    torch.manual_seed(5)
    batch = make_batch(64, seed=8, device=dev)
    batch.y = (torch.arange(64, device=dev) % 3 == 0).long()
    eager = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev)
    graphed = copy.deepcopy(eager)
    opt_e = configure_optimizer(eager, lr=1e-2, fused=True, capturable=True)
    opt_g = configure_optimizer(graphed, lr=1e-2, fused=True, capturable=True)
    with torch.no_grad():
        eager(batch); graphed(batch)                     # index plans are part of the resident input (built once, outside)
    # the forwards above ran batch norm in training mode on both models alike

    def step(model, opt):
        model.zero_grad(set_to_none=True)
        loss = model.loss(batch)
        loss.backward()
        opt.step()
        return loss

    losses_e = [float(step(eager, opt_e).detach()) for _ in range(4)]    # 2 warm-up-equivalent + 2 replayed-equivalent steps
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(graphed, opt_g)
        graphed.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            static_loss = graphed.loss(batch)
            static_loss.backward()
            opt_g.step()
    torch.cuda.current_stream().wait_stream(side)
    losses_g = []
    for _ in range(2):
        g.replay()
        losses_g.append(float(static_loss.detach()))
    torch.cuda.synchronize()
    # the capture itself does not execute: eager has done 4 steps, graphed 2 eager + 2 replayed
    assert losses_g == losses_e[2:], (losses_g, losses_e)
    for (name, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert torch.equal(pe, pg), name
    assert torch.equal(eager.gnn_model.node_batch_norm.running_var, graphed.gnn_model.node_batch_norm.running_var)
    assert int(eager.gnn_model.node_batch_norm.num_batches_tracked) == int(graphed.gnn_model.node_batch_norm.num_batches_tracked)


def _dp_rank(rank, world, port, out_dir):
    import os as _os
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from molkgnn_amd import dp
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    dev = torch.device("cuda:0")                          # both ranks share the one GPU of the test box
    assert dp.init_process_group_from_env("gloo") == world
    torch.manual_seed(11)                                 # identical replicas
    model = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev)
    names = [n for n, _ in model.named_parameters()]
    reducer = dp.FlatGradAllReduce(model.parameters(), dp.NEVER_TRAINED, names)

    def grads_of(seed):
        model.zero_grad(set_to_none=True)
        b = make_batch(48, seed=seed, device=dev)
        model.loss(b).backward()
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    mine = grads_of(300 + rank)
    reducer.reduce()
    reduced = {n: p.grad.clone().cpu() for n, p in model.named_parameters() if p.grad is not None}
    if rank == 0:
        other = grads_of(301)
        want = {n: ((mine[n] + other[n]) * 0.5).cpu() for n in mine}
        torch.save({"reduced": reduced, "want": want}, _os.path.join(out_dir, "dp_gpu.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_gradients_two_ranks_on_one_gpu(tmp_path):
    """Two ranks (gloo rendezvous, both on the test box's one GPU) run the HIP backward on different molecules; after the
    flat all-reduce every rank holds the average of the two gradients."""
    import socket
    import torch.multiprocessing as mp
    _dev()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = torch.load(os.path.join(str(tmp_path), "dp_gpu.pt"))
    assert len(res["want"]) > 40 and set(res["want"]) == set(res["reduced"])
    for n, w in res["want"].items():
        assert torch.allclose(res["reduced"][n], w, atol=1e-7, rtol=1e-6), n


def test_fixed_and_trainable_kernel_sets_compose_like_the_reference():
    """kernels.py:699-720: inside every degree block the fixed kernels' scores come first, then the trainable ones; a
    degree may have only one of the two.  Checked against the oracle run on each set separately and composed here."""
    from molkgnn_amd.kernels import BaseKernelSetConv, KernelConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(21)
    F, E = 28, 7
    mk = lambda L, d, grad: KernelConv(L=L, D=3, num_supports=d, node_attr_dim=F, edge_attr_dim=E, requires_grad=grad,
                                       init_support_attr_sc_weight=float(torch.rand(())), init_center_attr_sc_weight=float(torch.rand(())),
                                       init_edge_attr_support_sc_weight=float(torch.rand(())))
    fixed = [mk(3, 1, False), None, mk(2, 3, False), mk(4, 4, False)]         # degree 2: trainable only
    train = [mk(2, 1, True), mk(5, 2, True), None, mk(1, 4, True)]            # degree 3: fixed only
    layer = BaseKernelSetConv(*fixed, *train).to(dev)
    batch = make_batch(40, seed=12)
    bd = batch.to(dev)
    x = bd.x.clone().requires_grad_(True)
    out = layer._run(x, plan_from_data(bd), False)
    sd = lambda k: {n: p.detach().cpu() for n, p in k.named_parameters()}
    empty = lambda d: {"x_center": torch.zeros(0, F), "x_support": torch.zeros(0, d, F), "edge_attr_support": torch.zeros(0, d, E),
                       "p_support": torch.zeros(0, d, 3), "support_attr_sc_weight": torch.tensor(0.2),
                       "center_attr_sc_weight": torch.tensor(0.2), "edge_attr_support_sc_weight": torch.tensor(0.2)}
    per_f = [sd(k) if k is not None else empty(d + 1) for d, k in enumerate(fixed)]
    per_t = [sd(k) if k is not None else empty(d + 1) for d, k in enumerate(train)]
    xo = batch.x.clone().requires_grad_(True)
    of = O.kernelsetconv(per_f, xo, batch, False)
    ot = O.kernelsetconv(per_t, xo, batch, False)
    cols, a, c = [], 0, 0
    for d in range(4):
        nf = 0 if fixed[d] is None else fixed[d].num_kernels
        nt = 0 if train[d] is None else train[d].num_kernels
        cols += [of[:, a:a + nf], ot[:, c:c + nt]]
        a += nf; c += nt
    want = torch.cat(cols, dim=1)
    assert out.shape == want.shape == (batch.x.shape[0], 17)
    assert torch.allclose(out.detach().cpu(), want.detach(), atol=FWD_TOL, rtol=0)
    out.square().sum().backward()
    assert all(p.grad is None for k in fixed if k is not None for p in k.parameters() if p.dim() > 0)     # fixed banks stay fixed
    assert all(k.x_support.grad is not None and torch.isfinite(k.x_support.grad).all() for k in train if k is not None)
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    want.square().sum().backward()
    assert torch.allclose(x.grad.cpu(), xo.grad, atol=2e-5, rtol=1e-4), float((x.grad.cpu() - xo.grad).abs().max())


def test_very_wide_banks_fall_back_without_losing_columns():
    """100 kernels per degree need 22 (degree, column part) groups, the fused launch holds 16: the automatic variant must
    demote a degree to the generic kernels rather than drop column parts, and forcing the MFMA variant must say so."""
    from molkgnn_amd._lib import MolKGNNLibraryError
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    torch.manual_seed(2)
    b = make_batch(30, seed=44, device=dev)
    plan = plan_from_data(b)
    layer = KernelSetConv(100, 100, 100, 100, D=3, node_attr_dim=28, edge_attr_dim=7).to(dev)
    outs = {}
    for variant in ("generic", "auto"):
        layer.variant = variant
        x = b.x.clone().requires_grad_(True)
        o = layer._run(x, plan, False)
        o.square().sum().backward()
        outs[variant] = (o.detach(), x.grad.clone())
    assert torch.allclose(outs["generic"][0], outs["auto"][0], atol=FWD_TOL, rtol=0)
    assert torch.allclose(outs["generic"][1], outs["auto"][1], atol=5e-5, rtol=1e-3)
    deg = torch.bincount(b.edge_index[0], minlength=b.x.shape[0])
    for d in range(1, 5):                                    # every degree block carries scores for its atoms
        blk = outs["auto"][0][deg == d][:, 100 * (d - 1):100 * d]
        assert blk.numel() == 0 or float(blk.abs().min(dim=1).values.max()) > 0
    layer.variant = "mfma"
    with pytest.raises(MolKGNNLibraryError):
        layer._run(b.x, plan, False)


@pytest.mark.parametrize("F,E,Ls", [(20, 12, (3, 4, 2, 3)), (150, 7, (2, 3, 2, 2)), (27, 3, (4, 4, 4, 4))])
def test_shapes_outside_the_fast_paths_match_the_oracle(F, E, Ls):
    """Bond width > 8, row width > 112 or odd: the automatic variant takes the one-wave-per-atom kernels (forward and / or
    backward) -- forward and the x-gradient against the oracle."""
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.receptive_field import GraphBatch, build_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    topo = make_batch(12, seed=F, with_receptive_fields=False)
    g = torch.Generator().manual_seed(F + E)
    n, m = topo.x.shape[0], topo.edge_index.shape[1]
    x = torch.randn(n, F, generator=g)
    ea = torch.rand(m // 2, E, generator=g).repeat_interleave(2, dim=0)
    fields = build_receptive_fields(x, topo.p, topo.edge_index, ea)
    cpu = GraphBatch(x=x, p=topo.p, edge_index=topo.edge_index, edge_attr=ea, batch=topo.batch, **fields)
    bd = cpu.to(dev)
    torch.manual_seed(F)
    layer = KernelSetConv(*Ls, D=3, node_attr_dim=F, edge_attr_dim=E)
    per_degree = O.kernelset_params({k: v.detach().clone() for k, v in layer.state_dict().items()})
    layer = layer.to(dev)
    xg = bd.x.clone().requires_grad_(True)
    out = layer._run(xg, plan_from_data(bd), False)
    xo = x.clone().requires_grad_(True)
    want = O.kernelsetconv(per_degree, xo, cpu, False)
    assert torch.allclose(out.detach().cpu(), want.detach(), atol=FWD_TOL, rtol=0), float((out.detach().cpu() - want).abs().max())
    cot = torch.randn(want.shape, generator=g)
    (out * cot.to(dev)).sum().backward()
    (want * cot).sum().backward()
    assert torch.allclose(xg.grad.cpu(), xo.grad, atol=2e-5, rtol=1e-4), float((xg.grad.cpu() - xo.grad).abs().max())


# ------------------------------------------------------------------------------------------ optimiser --
def _adamw_models(dev, seed=3):
    torch.manual_seed(seed)
    shapes = [(10, 110), (20, 2, 110), (30, 3, 7), (3,), (1,), (32, 110), (32,), (1500, 3), (1, 32)]
    mine = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    return mine, ref


def test_fused_adamw_matches_torch_adamw():
    """mkgnn_adamw_step against torch.optim.AdamW (single-tensor path): two groups (with / without weight decay, as
    model.py:368-385 builds them), 12 steps, one parameter without a gradient on some steps (its step counter must
    lag, as torch's does), a learning rate changed on the way."""
    from molkgnn_amd.optim import FusedAdamW
    dev = _dev()
    mine, ref = _adamw_models(dev)
    mk = lambda ps, cls, **kw: cls([{"params": ps[:4], "weight_decay": 0.0}, {"params": ps[4:], "weight_decay": 0.05}],   # noqa: E731
                                   lr=3e-3, betas=(0.9, 0.99), eps=1e-8, **kw)
    opt_m, opt_r = mk(mine, FusedAdamW), mk(ref, torch.optim.AdamW, foreach=False, fused=False)
    g = torch.Generator(device=dev).manual_seed(17)
    for it in range(12):
        for pm, pr in zip(mine, ref):
            grad = torch.randn(pm.shape, generator=g, device=dev) * (10.0 ** ((it % 5) - 2))
            pm.grad, pr.grad = grad.clone(), grad.clone()
        if it % 3 == 1:
            mine[2].grad = None; ref[2].grad = None
        if it == 6:
            for o in (opt_m, opt_r):
                for grp in o.param_groups:
                    grp["lr"] = 1e-3
        opt_m.step(); opt_r.step()
    for i, (pm, pr) in enumerate(zip(mine, ref)):
        torch.testing.assert_close(pm, pr, rtol=2e-5, atol=2e-6, msg=lambda m, i=i: f"param {i}: {m}")
        sm, sr = opt_m.state[pm], opt_r.state[pr]
        assert float(sm["step"]) == float(sr["step"]), i
        # (gradients reach 1e2 and the first moment cancels: absolute tolerance on that scale)
        torch.testing.assert_close(sm["exp_avg"], sr["exp_avg"], rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(sm["exp_avg_sq"], sr["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    assert float(opt_m.state[mine[2]]["step"]) == 8.0


def test_fused_adamw_state_dict_round_trip_and_device_lr():
    """state_dict -> load_state_dict (into a fresh optimiser, and from torch.optim.AdamW's own state) continues the same
    trajectory; a 0-dim CUDA tensor as lr is read at run time (what a captured step needs for a schedule); maximize."""
    from molkgnn_amd.optim import FusedAdamW
    dev = _dev()
    mine, ref = _adamw_models(dev, seed=4)
    lr_t = torch.tensor(2e-3, device=dev)
    opt_m = FusedAdamW(mine, lr=lr_t, weight_decay=0.01, maximize=True)
    opt_r = torch.optim.AdamW(ref, lr=2e-3, weight_decay=0.01, maximize=True, foreach=False)
    g = torch.Generator(device=dev).manual_seed(5)

    def both_step(om, orf, pm_list, pr_list):
        for pm, pr in zip(pm_list, pr_list):
            grad = torch.randn(pm.shape, generator=g, device=dev)
            pm.grad, pr.grad = grad.clone(), grad.clone()
        om.step(); orf.step()
    for _ in range(3):
        both_step(opt_m, opt_r, mine, ref)
    lr_t.fill_(5e-4)                                     # the scheduler's write; no optimiser call involved
    for grp in opt_r.param_groups:
        grp["lr"] = 5e-4
    both_step(opt_m, opt_r, mine, ref)
    for pm, pr in zip(mine, ref):
        torch.testing.assert_close(pm, pr, rtol=2e-5, atol=2e-6)
    sd = opt_m.state_dict()
    assert all("_packed" not in st for st in sd["state"].values())
    # (a) our state into a fresh FusedAdamW over copies of the parameters
    mine2 = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt_m2 = FusedAdamW(mine2, lr=5e-4, weight_decay=0.01, maximize=True)
    opt_m2.load_state_dict(sd)
    # (b) torch's state into a fresh FusedAdamW
    mine3 = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    opt_m3 = FusedAdamW(mine3, lr=5e-4, weight_decay=0.01, maximize=True)
    opt_m3.load_state_dict(opt_r.state_dict())
    for _ in range(2):
        grads = [torch.randn(p.shape, generator=g, device=dev) for p in mine]
        for plist in (mine, ref, mine2, mine3):
            for p, gr in zip(plist, grads):
                p.grad = gr.clone()
        opt_m.step(); opt_r.step(); opt_m2.step(); opt_m3.step()
    for pm, pr, p2, p3 in zip(mine, ref, mine2, mine3):
        torch.testing.assert_close(pm, pr, rtol=2e-5, atol=2e-6)
        assert torch.equal(pm, p2)
        torch.testing.assert_close(p3, pr, rtol=2e-5, atol=2e-6)
    assert float(opt_m3.state[mine3[0]]["step"]) == 6.0


def test_fused_adamw_refuses_cpu_parameters():
    from molkgnn_amd import _lib
    from molkgnn_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.randn(4))
    p.grad = torch.randn(4)
    with pytest.raises(_lib.MolKGNNLibraryError):
        FusedAdamW([p]).step()


def test_head_dropout_mask_is_consistent_between_forward_and_backward():
    """Dropout inside the head kernels (model.py:150,169 ahead of ffn): the backward regenerates the forward's mask from
    the saved {seed, offset}.  The mask is read off grad_emb (zero exactly where an element was dropped); with it the
    loss and every gradient must equal the PyTorch formula; the keep rate is 1 - p; the state advances by one per
    forward, a fixed seed reproduces the masks, and a replayed graph draws a new mask each time."""
    from molkgnn_amd import readout as R
    dev = _dev()
    torch.manual_seed(9)
    B, H, p = 3000, 32, 0.25
    emb = torch.randn(B, H, device=dev, requires_grad=True)
    ffn = torch.nn.Linear(H, 1).to(dev)
    with torch.no_grad():
        ffn.weight.abs_().add_(0.1)                       # no zero weight: grad_emb == 0 means "dropped"
    y = (torch.rand(B, device=dev) < 0.3).long()

    def run():
        emb.grad = None; ffn.zero_grad()
        loss = R.bce_head_loss(emb, ffn, y, dropout_p=p)
        loss.backward()
        return loss.detach().clone(), emb.grad.clone(), ffn.weight.grad.clone(), ffn.bias.grad.clone()

    R.reset_head_rng(dev, seed=1234)
    loss1, ge1, gw1, gb1 = run()
    assert int(R.head_rng_state(dev)[1]) == 1
    mask = (ge1 != 0).float()
    keep = float(mask.mean())
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / (B * H)) ** 0.5 + 1e-3, keep
    assert 0.5 < float(mask[:, 0].mean()) < 0.95 and 0.5 < float(mask[0].mean()) <= 1.0      # no row / column structure
    e2 = emb.detach().clone().requires_grad_(True)
    w2, b2 = ffn.weight.detach().clone().requires_grad_(True), ffn.bias.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(((e2 * mask / (1 - p)) @ w2.t() + b2).view(-1), y.float())
    ref.backward()
    torch.testing.assert_close(loss1, ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ge1, e2.grad, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(gw1, w2.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gb1, b2.grad, rtol=1e-4, atol=1e-6)
    _, ge2, _, _ = run()                                   # next offset: another mask
    assert int(R.head_rng_state(dev)[1]) == 2
    assert float(((ge2 != 0) != (ge1 != 0)).float().mean()) > 0.2
    R.reset_head_rng(dev, seed=1234)                       # same seed, offset 0: the first mask again
    loss3, ge3, gw3, _ = run()
    assert torch.equal(ge3, ge1) and torch.equal(loss3, loss1) and torch.equal(gw3, gw1)
    # p = 0 through the same entry points is the plain head
    emb.grad = None
    l0 = R.bce_head_loss(emb, ffn, y, dropout_p=0.0)
    torch.testing.assert_close(l0.detach(), torch.nn.functional.binary_cross_entropy_with_logits(
        ffn(emb.detach()).view(-1), y.float()), rtol=1e-5, atol=1e-6)
    R.reset_head_rng(dev)


def test_head_dropout_draws_a_new_mask_on_every_graph_replay():
    """The generator's offset lives on the device and is advanced by the forward kernel: a captured step replayed
    twice uses two different masks (a captured torch dropout needs two extra fill kernels per replay for the same).
    (Its own test function on purpose: with any un-freed autograd graph over the same leaf alive at capture time --
    pure-PyTorch ones included -- ``capture_end`` of this PyTorch/ROCm build segfaults.)"""
    from molkgnn_amd import readout as R
    dev = _dev()
    torch.manual_seed(10)
    B, H, p = 3000, 32, 0.25
    emb = torch.randn(B, H, device=dev, requires_grad=True)
    ffn = torch.nn.Linear(H, 1).to(dev)
    with torch.no_grad():
        ffn.weight.abs_().add_(0.1)
    y = (torch.rand(B, device=dev) < 0.3).long()

    def run():
        emb.grad = None; ffn.zero_grad(set_to_none=True)
        R.bce_head_loss(emb, ffn, y, dropout_p=p).backward()

    R.reset_head_rng(dev, seed=77)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
        emb.grad = None; ffn.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            static_loss = R.bce_head_loss(emb, ffn, y, dropout_p=p)
            static_loss.backward()
    torch.cuda.current_stream().wait_stream(side)
    before = int(R.head_rng_state(dev)[1])
    g.replay(); a_mask = (emb.grad != 0).clone()
    g.replay(); b_mask = (emb.grad != 0).clone()
    torch.cuda.synchronize()
    assert int(R.head_rng_state(dev)[1]) == before + 2
    assert float((a_mask != b_mask).float().mean()) > 0.2
    R.reset_head_rng(dev)


# ------------------------------------------------------------------------------- block-row propagate --
def test_block_row_propagate_matches_dense_bitwise():
    """mkgnn_segment_sum_block_rows against the dense sum: (1) sources as block rows, with NaN everywhere outside the
    blocks (nothing there may be read) -> bit-identical h and row norms; (2) destinations as block rows -> every atom's
    own block bit-identical, the rest of the row untouched.  Atoms in no bucket (degree 0 / 9) contribute nothing."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.receptive_field import attach_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    for Ls in [(10, 20, 30, 50), (3, 0, 7, 6), (16, 16, 16, 16)]:
        b = make_batch(40, seed=21, device=dev)
        # a hub of degree 9 and an isolated atom: in no bucket
        n = b.x.shape[0]
        hub, extra = 0, torch.arange(n - 9, n - 1, device=dev)
        ei = torch.cat([b.edge_index, torch.stack([torch.full_like(extra, hub), extra]), torch.stack([extra, torch.full_like(extra, hub)])], dim=1)
        keep = (ei[0] != n - 1) & (ei[1] != n - 1)
        b.edge_index = ei[:, keep]
        b.edge_attr = torch.rand(b.edge_index.shape[1], 7, device=dev)
        attach_receptive_fields(b)
        plan = plan_from_data(b)
        K = sum(Ls)
        offs = [0, Ls[0], Ls[0] + Ls[1], Ls[0] + Ls[1] + Ls[2]]
        deg = plan.deg8.long()
        assert int((deg == 0).sum()) >= 2
        col = torch.arange(K, device=dev)[None, :]
        lo = torch.tensor([0] + offs, device=dev)[deg][:, None]
        ln = torch.tensor([0] + list(Ls), device=dev)[deg][:, None]
        inblock = (col >= lo) & (col < lo + ln)
        pad = (-K) % 4
        dense = torch.zeros(n, K + pad, device=dev)
        dense[:, :K] = torch.where(inblock, torch.randn(n, K, device=dev), torch.zeros((), device=dev))
        sparse = torch.full((n, K + pad), float("nan"), device=dev)
        sparse[:, :K] = torch.where(inblock, dense[:, :K], sparse[:, :K])
        h_ref = Fn.propagate_add(dense[:, :K], plan, out_pad=pad)
        sp = sparse[:, :K]
        setattr(sp, Fn._BLOCKS_ATTR, tuple(Ls))
        h_blk = Fn.propagate_add(sp, plan, out_pad=pad)
        assert torch.equal(h_ref, h_blk)
        assert torch.equal(getattr(h_ref, Fn._INV_ATTR)[0], getattr(h_blk, Fn._INV_ATTR)[0])
        # backward
        g = torch.randn(n, K + pad, device=dev)[:, :K]
        g_ref = Fn._segment_sum(g, plan.csr_out, pad)
        g_blk = Fn._segment_sum_blocks(g, plan.csr_out, plan.deg8, tuple(Ls), 2, pad, None)
        assert torch.equal(torch.where(inblock, g_ref, torch.zeros((), device=dev)),
                           torch.where(inblock, g_blk, torch.zeros((), device=dev)))


def test_molgcn_block_rows_equals_dense_path(monkeypatch):
    """The 3-layer MolGCN with sim_sc kept as block rows between convolution and propagate (the default) against the
    dense form (zero-filled rows, dense sums; MKGNN_DENSE_PROPAGATE): output and every gradient bit for bit.  (h between the
    layers as fp32 rows in both: the pre-split rows of round 6 -- the block-row path's default, tests/test_rows_split.py --
    change the row the backward gather un-normalises with by 2^-22.)"""
    from molkgnn_amd import KernelLayer as _KL
    monkeypatch.setattr(_KL, "_ROWS_SPLIT", False)
    import copy
    from molkgnn_amd import KernelLayer as KL
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    dev = _dev()
    torch.manual_seed(3)
    b = make_batch(64, seed=5, device=dev)
    net = MolKGNNNet(num_layers=3, num_kernel1_1hop=10, num_kernel2_1hop=20, num_kernel3_1hop=30, num_kernel4_1hop=50,
                     num_kernel1_Nhop=10, num_kernel2_Nhop=20, num_kernel3_Nhop=30, num_kernel4_Nhop=50,
                     x_dim=28, edge_attr_dim=7, graph_embedding_dim=32, drop_ratio=0.0).to(dev)
    net2 = copy.deepcopy(net)
    cot = torch.randn(64, 32, device=dev)
    res = []
    for model, blk in ((net, True), (net2, False)):
        old = KL._BLOCK_ROWS
        KL._BLOCK_ROWS = blk
        try:
            out = model(b)
            (out * cot).sum().backward()
        finally:
            KL._BLOCK_ROWS = old
        res.append((out.detach(), {n_: p.grad for n_, p in model.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) > 20
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_batch_norm_hands_row_norms_and_counts_batches():
    """mkgnn_batchnorm_forward also emits 1 / max(||row||, eps) of its output (28 channels: the first convolution reads
    it next) -- bit-identical to mkgnn_row_inv_norm on the same rows, so tie-breaks do not depend on who computed the
    norms -- and increments num_batches_tracked inside the kernel (no separate PyTorch kernel in the captured step)."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd import readout as R
    dev = _dev()
    torch.manual_seed(12)
    for n in (1, 7, 1000, 4097):
        if n == 1:
            continue                                      # BatchNorm1d refuses a single row in training mode
        bn = torch.nn.BatchNorm1d(28).to(dev).train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        x = torch.randn(n, 28, device=dev) * 3 + 1
        out = R.batch_norm(x, bn)
        assert int(bn.num_batches_tracked) == 1
        handed = Fn._handed_inv_norm(out)
        assert handed is not None and handed.shape == (n,)
        assert torch.equal(handed, Fn.row_inv_norm(out.detach()))
        ref = torch.nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.1, bn.eps)
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        R.batch_norm(x, bn)
        assert int(bn.num_batches_tracked) == 2
    bn = torch.nn.BatchNorm1d(40).to(dev).train()         # other widths: no norms handed over, same results
    out = R.batch_norm(torch.randn(50, 40, device=dev), bn)
    assert Fn._handed_inv_norm(out) is None and int(bn.num_batches_tracked) == 1


def test_fused_adamw_grad_scale_is_a_scaled_gradient():
    """grad_scale = 1 / world (the data-parallel step sums the gradients into the flat buffer and lets the optimiser
    divide): the same trajectory as torch.optim.AdamW fed with the scaled gradients."""
    from molkgnn_amd.optim import FusedAdamW
    dev = _dev()
    mine, ref = _adamw_models(dev, seed=6)
    opt_m = FusedAdamW(mine, lr=2e-3, weight_decay=0.02, grad_scale=0.125)
    opt_r = torch.optim.AdamW(ref, lr=2e-3, weight_decay=0.02, foreach=False)
    g = torch.Generator(device=dev).manual_seed(8)
    for _ in range(5):
        for pm, pr in zip(mine, ref):
            grad = torch.randn(pm.shape, generator=g, device=dev) * 8
            pm.grad, pr.grad = grad.clone(), grad * 0.125
        opt_m.step(); opt_r.step()
    for pm, pr in zip(mine, ref):
        torch.testing.assert_close(pm, pr, rtol=2e-5, atol=2e-6)


def test_receptive_field_builder_hip_matches_the_reference_transform():
    """G10 (the reference's own ``ToXAndPAndEdgeAttrForDeg`` outputs, collated): mkgnn_rf_count / mkgnn_rf_fill
    reproduce all 20 tensors exactly."""
    from molkgnn_amd.receptive_field import build_receptive_fields_hip
    dev = _dev()
    z = np.load(os.path.join(G.GOLDEN, "g10_receptive_fields.npz"))
    t = lambda k: torch.from_numpy(z[k])                  # noqa: E731
    got = build_receptive_fields_hip(t("batch/x").to(dev), t("batch/p").to(dev), t("batch/edge_index").to(dev),
                                     t("batch/edge_attr").to(dev))
    for d in range(1, 5):
        for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index"):
            k = f"{nm}_deg{d}"
            want = t("batch/" + k)
            assert got[k].numel() == want.numel(), k
            assert torch.equal(got[k].cpu().reshape(want.shape), want.to(got[k].dtype)), k


def test_plan_builder_hip_matches_torch_builder():
    """mkgnn_plan_build (scatter CSR of the backward, both propagate CSRs, deg8, packed columns) against the torch
    definition in plan.py (stable sorts): exact equality, on a synthetic batch, with the bonds shuffled, with hub atoms of
    degree 9 / isolated atoms / an absent degree, and for a plan without an edge list."""
    from molkgnn_amd.plan import plan_from_lists
    from molkgnn_amd.receptive_field import build_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    b = make_batch(300, seed=4, device=dev, with_receptive_fields=False)
    cases = [(b.x, b.p, b.edge_index, b.edge_attr)]
    g = torch.Generator().manual_seed(1)
    nb = b.edge_index.shape[1] // 2
    perm = torch.randperm(nb, generator=g).to(dev)
    eperm = torch.stack([2 * perm, 2 * perm + 1], dim=1).reshape(-1)
    cases.append((b.x, b.p, b.edge_index[:, eperm].contiguous(), b.edge_attr[eperm].contiguous()))
    pairs = [(i, 9) for i in range(9)] + [(12, 13), (13, 14)] + [(i, 24) for i in range(15, 24)] + [(24, 25)]
    ei = torch.tensor([[a, c] for a, c in pairs for (a, c) in ((a, c), (c, a))]).t().contiguous().to(dev)
    ea = torch.rand(ei.shape[1] // 2, 7, generator=g).repeat_interleave(2, dim=0).to(dev)
    cases.append((torch.randn(26, 28, device=dev), torch.randn(26, 3, device=dev), ei, ea))
    for ci, (x, p, edge_index, edge_attr) in enumerate(cases):
        f = build_receptive_fields(x, p, edge_index, edge_attr)
        lists = [[f[f"{nm}_deg{d}"] for d in range(1, 5)] for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index")]
        for with_edges in (True, False):
            hip = plan_from_lists(x.shape[0], *lists, edge_index if with_edges else None)
            assert hip.build_hip()
            ref = plan_from_lists(x.shape[0], *lists, edge_index if with_edges else None)
            os.environ["MKGNN_TORCH_PLAN"] = "1"
            try:
                want = [ref.scatter, ref.deg8] + ([ref.csr_in, ref.csr_out, ref.csr_in_packed] if with_edges else [])
            finally:
                del os.environ["MKGNN_TORCH_PLAN"]
            got = [hip.scatter, hip.deg8] + ([hip.csr_in, hip.csr_out, hip.csr_in_packed] if with_edges else [])
            for gi, (gv, wv) in enumerate(zip(got, want)):
                if isinstance(gv, tuple):
                    assert torch.equal(gv[0], wv[0]) and torch.equal(gv[1], wv[1]), (ci, with_edges, gi)
                else:
                    assert torch.equal(gv, wv), (ci, with_edges, gi)


def test_plan_dropped_before_anything_joins_the_index_stream():
    """``BatchPlan.build_hip`` runs its kernels on the device's index stream.  A plan (and its batch) dropped before anything
    read it used to hand its workspace back to the allocator of the CALLER's stream while those kernels were still running --
    the next batch's builder then wrote over it (a GPU memory fault, once, in tools/shard_loader_probe.py with six loader
    threads).  The buffers are now recorded on the index stream: build, drop, and at once allocate and overwrite blocks of
    the same sizes, many times over; the plan built last still equals the torch definition."""
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    base = make_batch(1500, seed=21, device=dev)
    names = [k for k, v in base.__dict__.items() if torch.is_tensor(v)]
    def clone():
        b = type(base)()
        for k, v in base.__dict__.items():
            if not k.startswith("_"):
                setattr(b, k, v.clone() if torch.is_tensor(v) else v)
        return b

    for it in range(24):
        b = clone()
        plan = plan_from_data(b)
        assert plan.build_hip()
        del plan, b                                              # nothing joined the index stream
        junk = [torch.full_like(getattr(base, k), -7 if not getattr(base, k).is_floating_point() else float("nan")) for k in names]
        junk += [torch.full((getattr(base, "x").shape[0] * 16,), -7, dtype=torch.int32, device=dev) for _ in range(6)]
        del junk
    hip = plan_from_data(clone())
    assert hip.build_hip()
    ref = plan_from_data(clone())                               # (another object: the plan is cached on the batch)
    os.environ["MKGNN_TORCH_PLAN"] = "1"
    try:
        want = [ref.scatter, ref.csr_in, ref.csr_out, ref.csr_in_packed]
    finally:
        del os.environ["MKGNN_TORCH_PLAN"]
    for gv, wv in zip([hip.scatter, hip.csr_in, hip.csr_out, hip.csr_in_packed], want):
        assert torch.equal(gv[0], wv[0]) and torch.equal(gv[1], wv[1])
    assert torch.equal(hip.deg8, ref.deg8)
    torch.cuda.synchronize()


def test_evaluation_metrics_on_the_device_match_reference_golden():
    """SURVEY 8 f-4 on the GPU: ``molkgnn_amd.evaluation`` (the reference's ``evaluation.py:11-127``: logAUC with
    scikit-learn's ``roc_curve`` / ``drop_intermediate`` and ``np.interp`` semantics, AUC, PPV, accuracy, F1) evaluated on
    CUDA tensors -- the validation scores never leave the device -- against the numbers the reference's own functions
    produced (tests/golden/g8_metrics.npz), to 1e-12; and on the scores of a model's forward on the device."""
    import os
    from molkgnn_amd import evaluation as E
    dev = _dev()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_metrics.npz"))
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/y")})
    assert len(names) == 6
    for name in names:
        y, s = torch.from_numpy(g[f"{name}/y"]).to(dev), torch.from_numpy(g[f"{name}/score"]).to(dev)
        for key, got in (("logauc", E.calculate_logAUC(y, s)), ("logauc_wide", E.calculate_logAUC(y, s, FPR_range=(0.01, 0.5))),
                         ("auc", E.calculate_auc(y, s)), ("ppv", E.calculate_ppv(y, s)), ("ppv_cut", E.calculate_ppv(y, s, cutoff=0.8)),
                         ("accuracy", E.calculate_accuracy(y, s)), ("f1", E.calculate_f1_score(y, s))):
            want = float(g[f"{name}/{key}"])
            if want != want:
                assert got != got, (name, key)
            else:
                assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (name, key, got, want)
    # scores straight from the HIP model: same metrics on the device and on the host copy
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    torch.manual_seed(8)
    b = make_batch(400, seed=80).to(dev)
    b.y = (torch.arange(400, device=dev) % 7 == 0).long()
    model = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev).eval()
    with torch.no_grad():
        pred, _ = model(b)
    score = torch.sigmoid(pred.view(-1))
    for fn in (E.calculate_logAUC, E.calculate_auc, E.calculate_ppv, E.calculate_accuracy, E.calculate_f1_score):
        a, c = fn(b.y, score), fn(b.y.cpu(), score.cpu())
        assert abs(a - c) <= 1e-12 * max(1.0, abs(c)), (fn.__name__, a, c)


@pytest.mark.parametrize("n,C,masked", [(300, 7, False), (300, 7, True), (218_000, 7, False), (218_000, 7, True), (5000, 1, False),
                                        (40_000, 33, False), (1, 7, False)])
def test_statistics_only_batch_norm_matches_torch(n, C, masked):
    """``readout.update_running_stats`` (mkgnn_batchnorm_update_stats): what ``bn(x)`` does to a BatchNorm1d's buffers in
    training mode without the normalised rows -- the reference's ``edge_batch_norm(data.edge_attr)`` (MolKGNNNet.py:116),
    whose output nothing reads.  One-launch and three-launch forms, the row mask of padded batches, three calls in a row."""
    from molkgnn_amd import readout as R
    dev = _dev()
    torch.manual_seed(n + C)
    mine, ref = torch.nn.BatchNorm1d(C).to(dev), torch.nn.BatchNorm1d(C).to(dev)
    for step in range(3):
        x = torch.randn(n, C, device=dev) * (1.0 + step) + 0.3 * step
        key = lim = None
        xr = x
        if masked:
            key = torch.randint(0, 1000, (n,), device=dev)
            lim = torch.tensor([700], dtype=torch.int64, device=dev)
            xr = x[key < 700]
        R.update_running_stats(x, mine, key, lim)
        if xr.shape[0] > 1:
            ref(xr)
        # (torch RAISES on one value per channel in training mode; a kernel cannot, so such a degenerate batch leaves the
        # statistics and the counter where they are -- kgnn_readout.hip bn_side_final)
        assert torch.allclose(mine.running_mean, ref.running_mean, atol=2e-6, rtol=1e-5), step
        assert torch.allclose(mine.running_var, ref.running_var, atol=1e-5, rtol=2e-5), step
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == (step + 1 if n > 1 else 0)
    mine.eval()
    before, count = mine.running_mean.clone(), int(mine.num_batches_tracked)
    R.update_running_stats(torch.randn(n, C, device=dev), mine)
    assert torch.equal(mine.running_mean, before) and int(mine.num_batches_tracked) == count      # eval mode: nothing moves


@pytest.mark.parametrize("mols,padded", [(64, False), (700, False), (300, True)])
def test_every_buffer_of_the_state_dict_after_three_training_steps(mols, padded):
    """state-dict contents are API.  After three training steps every BUFFER of ``MolKGNNNet`` -- node_batch_norm's and
    edge_batch_norm's running_mean / running_var / num_batches_tracked -- equals what torch.nn.BatchNorm1d leaves when fed
    the same rows (reference MolKGNNNet.py:115-116: both batch norms run in every forward).  Padded batches
    (molkgnn_amd.padding) count their real atoms and the bonds of their real atoms only."""
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd import padding as P
    dev = _dev()
    torch.manual_seed(3)
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    model = MolKGNNNet(num_layers=2, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32,
                       **dict(zip(names, (10, 20, 30, 50) * 2))).to(dev).train()
    ref_node, ref_edge = torch.nn.BatchNorm1d(28).to(dev), torch.nn.BatchNorm1d(7).to(dev)
    raws = [make_batch(mols, seed=900 + s, with_receptive_fields=not padded) for s in range(3)]
    if padded:
        shape = P.fixed_shape([P.degree_histogram(b) for b in raws])
    for raw in raws:
        if padded:
            from molkgnn_amd.receptive_field import attach_receptive_fields
            b = P.pad_batch(raw, shape, mols).to(dev)
            attach_receptive_fields(b, sizes=b.bucket_sizes)
            assert b.edge_attr.shape[0] > raw.edge_attr.shape[0]                  # (there are padding bonds to leave out)
        else:
            b = raw.to(dev)
        model(b).sum().backward()
        ref_node(raw.x.to(dev)); ref_edge(raw.edge_attr.to(dev))
    torch.cuda.synchronize()
    got = dict(model.named_buffers())
    assert sorted(got) == ["edge_batch_norm.num_batches_tracked", "edge_batch_norm.running_mean", "edge_batch_norm.running_var",
                           "node_batch_norm.num_batches_tracked", "node_batch_norm.running_mean", "node_batch_norm.running_var"]
    for pre, ref in (("node_batch_norm", ref_node), ("edge_batch_norm", ref_edge)):
        assert torch.allclose(got[f"{pre}.running_mean"], ref.running_mean, atol=2e-6, rtol=1e-5), pre
        assert torch.allclose(got[f"{pre}.running_var"], ref.running_var, atol=1e-5, rtol=2e-5), pre
        assert int(got[f"{pre}.num_batches_tracked"]) == int(ref.num_batches_tracked) == 3, pre
    model.eval()
    with torch.no_grad():
        model(b)
    assert int(model.edge_batch_norm.num_batches_tracked) == 3                                  # eval mode moves nothing


def test_bench_two_ranks_on_one_card_reports_the_data_parallel_keys(tmp_path):
    """``python bench.py --gpus 2`` through its own launcher, both ranks on the test box's one GPU (gloo): the N > 1 line
    must explain itself -- per-rank step times, the all-reduce's own time, the step form, parameters AND buffers
    identical over the ranks after the run."""
    import json
    import subprocess
    import sys
    _dev()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MKGNN_ALLOW_SHARED_GPU="1", MKGNN_DIST_BACKEND="gloo", MKGNN_NO_SMALL_BATCH="1", MKGNN_NO_SHARD_EPOCH="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "2",
                        "--batch-size", "256", "--fresh-batches", "0", "--no-cpu-baseline", "--roofline-reps", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dp_ranks_seen"] == 2 and line["graph_replay"] is True
    assert line["dp_one_graph"] is False and "optimiser graph" in line["dp_step"]
    assert line["dp_replicas_max_abs_diff"] == 0.0
    pr = line["dp_per_rank"]
    assert len(pr["ms_per_step_by_rank"]) == 2 and pr["ms_per_step_min"] <= pr["ms_per_step_max"] <= line["ms_per_step"] * 1.001 + 1e-3
    ar = line["dp_allreduce_ms"]
    assert ar["median"] > 0 and ar["bytes"] > 4 * 120_000
    db = line["dp_buffers"]
    assert db["max_abs_diff_before_sync"] > 0 and db["max_abs_diff"] == 0.0      # the ranks' batches differ; averaged once at the end


def test_one_pass_index_builder_matches_the_separate_builders():
    """mkgnn_index_build (round 5: receptive fields AND index plan from one memset + six kernels) against the definitions:
    the 20 per-degree tensors + unit bond rows of the torch receptive-field builder (itself pinned to the reference's
    wrapper.py:559-672 by G10) and the torch plan builder's scatter CSR, propagate CSRs, deg8 and packed columns -- exact
    equality, on a synthetic batch, with the bonds shuffled, with hub atoms of degree 9 / isolated atoms / an absent degree,
    and on the reference's own G10 molecules."""
    from molkgnn_amd import _lib
    from molkgnn_amd.plan import plan_from_lists
    from molkgnn_amd.receptive_field import build_index_hip, build_receptive_fields, check_sizes
    from molkgnn_amd.synthetic import make_batch
    dev = _dev()
    b = make_batch(300, seed=4, device=dev, with_receptive_fields=False)
    cases = [(b.x, b.p, b.edge_index, b.edge_attr)]
    g = torch.Generator().manual_seed(1)
    nb = b.edge_index.shape[1] // 2
    perm = torch.randperm(nb, generator=g).to(dev)
    eperm = torch.stack([2 * perm, 2 * perm + 1], dim=1).reshape(-1)
    cases.append((b.x, b.p, b.edge_index[:, eperm].contiguous(), b.edge_attr[eperm].contiguous()))
    pairs = [(i, 9) for i in range(9)] + [(12, 13), (13, 14)] + [(i, 24) for i in range(15, 24)] + [(24, 25)]
    ei = torch.tensor([[a, c] for a, c in pairs for (a, c) in ((a, c), (c, a))]).t().contiguous().to(dev)
    ea = torch.rand(ei.shape[1] // 2, 7, generator=g).repeat_interleave(2, dim=0).to(dev)
    cases.append((torch.randn(26, 28, device=dev), torch.randn(26, 3, device=dev), ei, ea))
    big = make_batch(4096, seed=77, device=dev, with_receptive_fields=False)       # many scan blocks (2048 atoms each)
    cases.append((big.x, big.p, big.edge_index, big.edge_attr))
    lib = _lib.load()
    for ci, (x, p, edge_index, edge_attr) in enumerate(cases):
        f = build_receptive_fields(x, p, edge_index, edge_attr)
        sizes = [int(f[f"selected_index_deg{d}"].numel()) for d in range(1, 5)]
        rf, parts = build_index_hip(x, p, edge_index, edge_attr, sizes)
        check_sizes(rf)
        assert rf["rf_counts"].tolist()[:5] == sizes + [0]
        for d in range(1, 5):
            for nm in ("selected_index", "nei_index", "p_focal", "nei_p", "nei_edge_attr"):
                got, want = rf[f"{nm}_deg{d}"], f[f"{nm}_deg{d}"]
                assert got.numel() == want.numel() and torch.equal(got.reshape(-1), want.reshape(-1)), (ci, nm, d)
            if sizes[d - 1]:
                unit = torch.empty((sizes[d - 1] * d, 8), dtype=torch.float32, device=dev)
                raw = f[f"nei_edge_attr_deg{d}"].reshape(-1, 7).contiguous()
                _lib.check(lib.mkgnn_unit_rows8(raw.data_ptr(), raw.shape[0], 7, unit.data_ptr(), _lib.stream_ptr(dev)), "unit")
                assert torch.equal(rf[f"nei_edge_unit_deg{d}"], unit), (ci, d)
        lists = [[f[f"{nm}_deg{d}"] for d in range(1, 5)] for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index")]
        ref = plan_from_lists(x.shape[0], *lists, edge_index)
        os.environ["MKGNN_TORCH_PLAN"] = "1"
        try:
            want = [ref.scatter, ref.deg8, ref.csr_in, ref.csr_out, ref.csr_in_packed]
        finally:
            del os.environ["MKGNN_TORCH_PLAN"]
        got = [parts["scatter"], parts["deg8"], parts["csr_in"], parts["csr_out"], parts["csr_in_packed"]]
        for gi, (gv, wv) in enumerate(zip(got, want)):
            if isinstance(gv, tuple):
                assert torch.equal(gv[0], wv[0]) and torch.equal(gv[1], wv[1]), (ci, gi)
            else:
                assert torch.equal(gv, wv), (ci, gi)
    # capacities that do not match the batch: reported, not silent
    x, p, edge_index, edge_attr = cases[0]
    f = build_receptive_fields(x, p, edge_index, edge_attr)
    sizes = [int(f[f"selected_index_deg{d}"].numel()) for d in range(1, 5)]
    small = [sizes[0], sizes[1] - 3, sizes[2], sizes[3]]
    rf, _ = build_index_hip(x, p, edge_index, edge_attr, small)
    with pytest.raises(ValueError):
        check_sizes(rf)


@pytest.mark.parametrize("overlap", [False, True])
def test_training_step_with_the_one_pass_index_builder(overlap, monkeypatch):
    """MKGNN_MERGED_INDEX=1: a padded batch's receptive fields and index plan from mkgnn_index_build (attached to the plan
    cache, the first convolution waiting on the event recorded inside the call) -- loss and every gradient bit for bit those
    of the same step with the separate builders."""
    from molkgnn_amd import padding as P
    from molkgnn_amd.receptive_field import attach_receptive_fields, check_sizes
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    dev = _dev()
    torch.manual_seed(21)
    model = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev).train()
    raws = [make_batch(200, seed=70 + i, with_receptive_fields=False) for i in range(2)]
    for i, r in enumerate(raws):
        r.y = ((torch.arange(200) + i) % 3 == 0).long()
    shape = P.fixed_shape([P.degree_histogram(r) for r in raws])
    results = {}
    for merged in ("0", "1"):
        monkeypatch.setenv("MKGNN_MERGED_INDEX", merged)
        out = []
        for r in raws:
            pb = P.pad_batch(r, shape, 200).to(dev)
            attach_receptive_fields(pb, sizes=pb.bucket_sizes, overlap=overlap)
            model.zero_grad(set_to_none=True)
            loss = model.loss(pb)
            loss.backward()
            torch.cuda.synchronize()
            check_sizes(pb)
            out.append((loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        results[merged] = out
    for (l0, g0), (l1, g1) in zip(results["0"], results["1"]):
        assert torch.equal(l0, l1)
        assert g0.keys() == g1.keys() and all(torch.equal(g0[k], g1[k]) for k in g0)


def test_touch_hint_is_taken_once_and_changes_nothing(monkeypatch):
    """``mkgnn_touch_hint`` (ABI v7, ``functional.touch_hint``): the batch's index arrays are read by spare blocks of the batch norm's
    statistics launch (training mode) or of the bank preparation -- by whichever comes first, once; a hint nobody takes is
    withdrawn when the block is left; and no result depends on it: a training step's loss and gradients are bit for bit those of
    ``MKGNN_TOUCH=0``."""
    from molkgnn_amd import functional as Fn, readout as R
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, backward as train_backward
    dev = _dev()
    b = make_batch(300, seed=5).to(dev)
    plan = plan_from_data(b)
    arrays = Fn.plan_touch_list(plan)
    assert 4 <= len(arrays) <= 16 and all(t.is_cuda for t in arrays)
    bn = torch.nn.BatchNorm1d(28).to(dev)
    with Fn.touch_hint(arrays) as h:                      # training mode: the statistics launch takes it
        R.batch_norm(b.x, bn)
    assert h.taken
    bn.eval()
    with Fn.touch_hint(arrays) as h:                      # eval mode: no statistics launch, nobody takes it -- withdrawn
        R.batch_norm(b.x, bn)
    assert not h.taken
    with Fn.touch_hint(arrays) as h:                      # ... and a bank preparation does
        layer = GNNModel().to(dev).gnn_model.gnn.layers[0]
        params, E = layer._bank_params("train", b.x)
        Fn.prepare_banks([params], [28], E, b.x.shape[0], plan.n_slots)
    assert h.taken
    with Fn.touch_hint([]) as h:
        pass
    assert not h.taken
    torch.cuda.synchronize()

    def step(touch):
        monkeypatch.setattr(Fn, "TOUCH", touch)
        torch.manual_seed(11)
        model = GNNModel(ffn_dropout_rate=0.0).to(dev)
        model.train()
        loss = model.loss(b)
        train_backward(loss)
        torch.cuda.synchronize()
        return float(loss), [p.grad.clone() for p in model.parameters() if p.grad is not None]
    l1, g1 = step(True)
    l0, g0 = step(False)
    assert l1 == l0 and len(g1) == len(g0) and all(torch.equal(a, c) for a, c in zip(g1, g0))


def test_deferred_bank_preparation_is_carried_by_the_batch_norm_and_changes_nothing(monkeypatch):
    """``mkgnn_bank_prepare_deferred`` (ABI v7, ``functional.prepare_banks(defer=True)``): the preparation left pending is carried by
    blocks of the next training-mode batch norm's statistics launch (``mkgnn_bank_prepare_flush`` then finds nothing), or launched by
    the flush (eval-mode batch norm: no statistics launch); either way the prepared workspaces are byte for byte those of the
    immediate launch, and a training step of the model is bit for bit the step with ``MKGNN_PREPARE_DEFER=0``."""
    from molkgnn_amd import KernelLayer as KL, functional as Fn, readout as R
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel, backward as train_backward
    dev = _dev()
    b = make_batch(300, seed=5).to(dev)
    plan = plan_from_data(b)
    torch.manual_seed(3)
    gnn = GNNModel().to(dev).gnn_model.gnn
    pl = [layer._bank_params("train", b.x) for layer in gnn.layers]
    Fs = [28, 110, 110]
    args = ([p for p, _ in pl], Fs, pl[0][1], b.x.shape[0], plan.n_slots)
    lib = R._lib.load()
    xs = [b.x] + [torch.randn(b.x.shape[0], 112, device=dev)[:, :110] for _ in range(2)]

    def outputs(prepared):
        """every layer's forward on the prepared banks (the kernels read nothing of the parameters but what the preparation wrote)"""
        with torch.no_grad():
            return [Fn._forward_impl(xs[k], plan, k == 2, Fn.VARIANTS["auto"], 0, pl[k][1], [p.detach() for p in pl[k][0]], False,
                                     None, prepared[k])[1].clone() for k in range(3)]
    want = outputs(Fn.prepare_banks(*args))
    for training in (True, False):
        bn = torch.nn.BatchNorm1d(28).to(dev)
        bn.train(training)
        got = Fn.prepare_banks(*args, defer=True)
        R.batch_norm(b.x, bn)
        pending = int(lib.mkgnn_bank_prepare_withdraw())
        assert pending == (0 if training else 1), (training, pending)
        if not training:                                  # nobody carried it: arm again and flush
            got = Fn.prepare_banks(*args, defer=True)
            Fn.prepare_flush(dev)
            assert int(lib.mkgnn_bank_prepare_withdraw()) == 0
        for k, (w, g) in enumerate(zip(want, outputs(got))):
            assert torch.equal(w, g), (training, k)

    def step(defer):
        monkeypatch.setattr(KL, "_PREPARE_DEFER", defer)
        torch.manual_seed(11)
        model = GNNModel(ffn_dropout_rate=0.0).to(dev)
        model.train()
        loss = model.loss(b)
        train_backward(loss)
        torch.cuda.synchronize()
        return float(loss), [p.grad.clone() for p in model.parameters() if p.grad is not None]
    l1, g1 = step(True)
    l0, g0 = step(False)
    assert l1 == l0 and len(g1) == len(g0) and all(torch.equal(a, c) for a, c in zip(g1, g0))


def test_flat_copy_fills_the_gradient_buffer_like_foreach_copy():
    """``mkgnn_flat_copy`` (ABI v7): ``dp.FlatGradAllReduce._fill`` -- the data-parallel step's gradients (and has-gradient flags) into
    the flat buffer one collective sums -- as ONE launch; the buffer is bit for bit ``torch._foreach_copy_``'s, tensors of 1 to
    5 500 elements, a missing gradient filled with zeros."""
    from molkgnn_amd import dp
    from molkgnn_amd.train import GNNModel
    dev = _dev()
    torch.manual_seed(5)
    model = GNNModel().to(dev)
    params = list(model.parameters())
    red = dp.FlatGradAllReduce(params, dp.NEVER_TRAINED, [n for n, _ in model.named_parameters()])
    grads = [torch.randn_like(p) for p in red.params]
    grads[3] = None
    red.prepare_patterns([grads])
    used = []
    orig = red._hip_copy
    red._hip_copy = lambda d, s: (used.append(1), orig(d, s))[1]
    red._fill(grads)
    torch.cuda.synchronize()
    assert used, "the HIP copy was not taken"
    got = red._buf.clone()
    red._buf.zero_()
    red._hip_copy = lambda d, s: False                    # the PyTorch path
    red._fill(grads)
    torch.cuda.synchronize()
    assert torch.equal(got, red._buf)
    assert float(red.flags.sum()) == len(grads) - 1
