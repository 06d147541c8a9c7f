"""How far is the drop-in's output from what the REFERENCE would have produced on the same inputs -- with nobody's
permutation choices forced?  

Every network-level parity test replays the oracle with the build's own neighbour orders (``forced_idx``): among
mathematically tied orders ``torch.max`` (reference ``kernels.py:373``) follows fp32 rounding, the build follows its fixed
rule, and either choice is right -- but a user who swaps the reference's kernels for this build gets the
UN-forced difference.  Reported per layer:

* the fraction of (atom, kernel) scores -- own-degree block only, the rest of a row is zero on both sides -- that differ
  from the reference's by more than 1e-5, and the largest difference;
* max |h - h_ref| of the layer's propagated output, absolute and relative to max |h_ref|;
* at the end max |embedding - reference embedding| (absolute, relative).

Cases: ``g3`` / ``g7`` -- the reference's OWN outputs from the golden files (tests/golden/make_golden.py: 3 molecules with
reduced banks, 4 molecules with the full-size seeded model); ``bench`` -- the benchmark's first batch (4 096 molecules, seed
1798000, bench.py) through the full-size seeded model against the oracle's reference-faithful form with its own argmax
(oracle/kgnn_oracle.py, pinned to the reference by the golden files).

    python -m tests.unforced_distance [g3 g7 bench] [--molecules 4096]

Test infrastructure (it runs the oracle): lives under tests/, imported by tests/test_scale_parity.py.
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ -> the repository root
if REPO not in sys.path:
    sys.path.insert(0, REPO)

TOL = 1e-5


def _build_layers(model, bd, train_bn):
    """The build's per-layer (sim_sc, h) and embedding, nothing forced: the model's own operators."""
    from molkgnn_amd import functional as Fn
    from molkgnn_amd import readout as R
    from molkgnn_amd.plan import plan_from_data
    plan = plan_from_data(bd)
    out = []
    with torch.no_grad():
        h = R.batch_norm(bd.x, model.node_batch_norm, None)
        n_layers = len(model.gnn.layers)
        for i, layer in enumerate(model.gnn.layers):
            params, E = layer._bank_params("train", h)
            sim, _ = Fn.kernelsetconv_details(h, plan, i == n_layers - 1, params, E)
            h = Fn.propagate_add(sim, plan, out_pad=(-sim.shape[1]) % 4)
            out.append((sim.cpu(), h.cpu()))
        emb = model(bd).cpu()
    return out, emb


def _compare(build, emb, ref_layers, ref_emb, batch, Ls):
    rows = []
    off = np.concatenate([[0], np.cumsum(Ls)])
    for i, ((sim, h), (sim_r, h_r)) in enumerate(zip(build, ref_layers)):
        n_pairs = n_bad = 0
        worst = 0.0
        for d in range(1, 5):
            sel = getattr(batch, f"selected_index_deg{d}")
            if sel.numel() == 0 or Ls[d - 1] == 0:
                continue
            blk = (sim[sel][:, off[d - 1]:off[d]] - sim_r[sel][:, off[d - 1]:off[d]]).abs()
            n_pairs += blk.numel()
            n_bad += int((blk > TOL).sum())
            worst = max(worst, float(blk.max()))
        dh = float((h - h_r).abs().max())
        rows.append({"layer": i, "pairs": n_pairs, "pairs_differing": n_bad, "fraction": n_bad / max(n_pairs, 1),
                     "max_score_diff": worst, "max_h_diff": dh, "max_h_diff_rel": dh / max(float(h_r.abs().max()), 1e-30)})
    de = float((emb - ref_emb).abs().max())
    return rows, {"max_embedding_diff": de, "max_embedding_diff_rel": de / max(float(ref_emb.abs().max()), 1e-30),
                  "embedding_scale": float(ref_emb.abs().max())}


def measure(case: str, molecules: int = 4096):
    """-> (per-layer rows, embedding summary) of one case; needs the GPU (the build's side) -- the reference side is the
    golden file or the CPU oracle."""
    from molkgnn_amd.MolKGNNNet import MolKGNNNet
    from molkgnn_amd.receptive_field import GraphBatch
    from molkgnn_amd.synthetic import make_batch
    from oracle import kgnn_oracle as O
    dev = torch.device("cuda:0")
    names = [f"num_kernel{d}_{h}" for h in ("1hop", "Nhop") for d in range(1, 5)]
    gold = os.path.join(REPO, "tests", "golden")
    if case == "g3":
        flat = dict(np.load(os.path.join(gold, "g3_molkgnnnet.npz")))
        kc = [int(v) for v in flat["kernel_counts"]]
        model = MolKGNNNet(num_layers=3, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32, **dict(zip(names, kc)))
        state = {k[len("param/"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("param/")}
        state.update({k[len("buffer/"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("buffer/")})
        model.load_state_dict(state, strict=True)
        b = GraphBatch(**{k[len("in_"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("in_")})
        b.num_graphs = 3
        ref_layers = [(torch.from_numpy(flat[f"layer{i}_sim_sc"]), torch.from_numpy(flat[f"layer{i}_h"])) for i in range(3)]
        ref_emb = torch.from_numpy(flat["graph_embedding"])
        Ls = kc[4:]
        model = model.to(dev).eval()
        build, emb = _build_layers(model, b.to(dev), False)
        # (layer 0 runs the 1-hop bank: kc[:4]; its widths are what the golden model was built with)
        rows, summ = _compare(build, emb, ref_layers, ref_emb, b, Ls if kc[:4] == kc[4:] else kc[:4])
        return rows, summ
    if case == "g7":
        flat = dict(np.load(os.path.join(gold, "g7_fullsize.npz")))
        torch.manual_seed(int(flat["seed"]))
        model = MolKGNNNet(num_layers=3, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32,
                           **dict(zip(names, (10, 20, 30, 50) * 2)))
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        b = GraphBatch(**{k[len("in_"):]: torch.from_numpy(v) for k, v in flat.items() if k.startswith("in_")})
        b.num_graphs = 4
        # the golden file holds the reference's layer-0 scores and embedding; the layers in between: the oracle's faithful form
        coll = []
        with torch.no_grad():
            O.molkgnnnet(state, b, 3, training_bn=False, form="faithful", collect=coll)
        ref_layers = [(s.detach(), h.detach()) for s, h in coll]
        assert torch.allclose(ref_layers[0][0], torch.from_numpy(flat["layer0_sim_sc"]), atol=1e-6)   # the oracle IS the reference here
        ref_emb = torch.from_numpy(flat["graph_embedding"])
        model = model.to(dev).eval()
        build, emb = _build_layers(model, b.to(dev), False)
        return _compare(build, emb, ref_layers, ref_emb, b, (10, 20, 30, 50))
    if case == "bench":
        torch.manual_seed(1798)
        model = MolKGNNNet(num_layers=3, x_dim=28, p_dim=3, edge_attr_dim=7, drop_ratio=0.0, graph_embedding_dim=32,
                           **dict(zip(names, (10, 20, 30, 50) * 2)))
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        b = make_batch(molecules, seed=1798000)
        b.num_graphs = molecules
        coll = []
        from bench import host_cores                     # (the cgroup's quota, not the affinity mask: 16 of the box's cores)
        torch.set_num_threads(host_cores())
        with torch.no_grad():
            ref_emb = O.molkgnnnet(state, b, 3, training_bn=False, form="faithful", collect=coll)
        ref_layers = [(s.detach(), h.detach()) for s, h in coll]
        model = model.to(dev).eval()
        build, emb = _build_layers(model, b.to(dev), False)
        return _compare(build, emb, ref_layers, ref_emb.detach(), b, (10, 20, 30, 50))
    raise ValueError(case)


def report(case, rows, summ):
    print(f"case {case}:")
    print("| layer | (atom, kernel) scores | differing by > 1e-5 | fraction | largest score difference | max |h - h_ref| | relative |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['layer']} | {r['pairs']} | {r['pairs_differing']} | {r['fraction']:.3e} | {r['max_score_diff']:.3e} | "
              f"{r['max_h_diff']:.3e} | {r['max_h_diff_rel']:.3e} |")
    print(f"embedding: max |build - reference| = {summ['max_embedding_diff']:.3e} (relative to max |reference| = "
          f"{summ['embedding_scale']:.3e}: {summ['max_embedding_diff_rel']:.3e})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["g3", "g7", "bench"])
    ap.add_argument("--molecules", type=int, default=4096)
    a = ap.parse_args()
    for c in a.cases:
        report(c, *measure(c, a.molecules))
