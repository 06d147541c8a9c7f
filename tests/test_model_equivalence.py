"""Model-level equivalence of the un-forced drop-in (missing 3).  ``pytest -m gpu``.

Layer by layer the build is NOT the reference "within 1e-5": from layer 1 on two neighbours of an atom can hold bit-identical
rows, ``torch.max`` (kernels.py:373) then follows fp32 rounding among mathematically tied neighbour orders, the build its fixed
rule, and where the tied neighbours carry different bond attributes the edge score differs (DESIGN section 2
measures 8.8 % of the layer-2 scores of the benchmark batch).  What CAN be asked of a drop-in is that a model TRAINED with it
behaves like one trained with the reference's path.  So: the same initial parameters, the same 80 batches of 16 molecules with a
label the network can learn (at least two degree-4 atoms, tools/train_synthetic.py's), the same AdamW --

  (a) the CPU oracle's reference-faithful form with ITS OWN argmax (nothing forced), autograd in PyTorch;
  (b) the build (HIP kernels, its own neighbour orders);
  (c) the build again from a second initialisation seed: what "run-to-run" looks like for this task;

and the loss trajectory of (b) must follow (a) far more closely than (c) follows (b), the held-out AUC / logAUC
(evaluation.py:11-127) of (a) and (b) must agree within the (b)-(c) spread.  The table is printed (``pytest -s``) and quoted in
DESIGN section 2.
"""
import time

import pytest
import torch

from oracle import kgnn_oracle as O

pytestmark = pytest.mark.gpu

STEPS, BATCH, HELD_OUT, LR = 80, 16, 384, 3e-3


def _labelled(n, seed):
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(n, seed=seed)
    deg = torch.bincount(b.edge_index[0], minlength=b.x.shape[0])
    n4 = torch.zeros(n).index_add_(0, b.batch, (deg == 4).float())
    b.y = (n4 >= 2).float()
    return b


def _oracle_run(init_state, train, held):
    """(a): plain PyTorch on the CPU -- oracle.molkgnnnet (faithful form, own argmax) + ffn + BCEWithLogits + torch AdamW."""
    state = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k)
             for k, v in init_state.items()}
    gstate = {k[len("gnn_model."):]: v for k, v in state.items() if k.startswith("gnn_model.")}
    leaves = [v for v in state.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=LR, weight_decay=0.0)
    losses = []
    for b in train:
        opt.zero_grad(set_to_none=True)
        emb = O.molkgnnnet(gstate, b, num_layers=3, training_bn=True, form="faithful")
        pred = emb @ state["ffn.weight"].T + state["ffn.bias"]
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred.view(-1), b.y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    with torch.no_grad():
        emb = O.molkgnnnet(gstate, held, num_layers=3, training_bn=False, form="faithful")
        pred = (emb @ state["ffn.weight"].T + state["ffn.bias"]).view(-1)
    return losses, pred


def _build_run(model, train_gpu, held_gpu):
    """(b), (c): the build -- GNNModel.loss (the path a run takes at this batch size), torch AdamW as for the oracle."""
    opt = torch.optim.AdamW([p for p in model.parameters()], lr=LR, weight_decay=0.0)
    losses = []
    model.train()
    for b in train_gpu:
        opt.zero_grad(set_to_none=True)
        loss = model.loss(b)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    model.eval()
    with torch.no_grad():
        pred, _ = model(held_gpu)
    return losses, pred.view(-1).cpu()


def test_a_model_trained_with_the_build_behaves_like_one_trained_with_the_reference_path(capsys):
    from molkgnn_amd import evaluation as E
    from molkgnn_amd.train import GNNModel
    assert torch.cuda.is_available(), "pytest -m gpu needs the MI355X"
    dev = torch.device("cuda:0")
    train = [_labelled(BATCH, 7000 + i) for i in range(STEPS)]
    held = _labelled(HELD_OUT, 9999)
    train_gpu, held_gpu = [b.to(dev) for b in train], held.to(dev)
    torch.manual_seed(1798)
    model_b = GNNModel(ffn_dropout_rate=0.0)
    init = {k: v.detach().clone() for k, v in model_b.state_dict().items()}
    torch.manual_seed(4242)
    model_c = GNNModel(ffn_dropout_rate=0.0)
    t0 = time.perf_counter()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    la, pa = _oracle_run(init, train, held)
    t_oracle = time.perf_counter() - t0
    lb, pb = _build_run(model_b.to(dev), train_gpu, held_gpu)
    lc, pc = _build_run(model_c.to(dev), train_gpu, held_gpu)
    y = held.y
    met = {k: (float(E.calculate_auc(y, p)), float(E.calculate_logAUC(y, p))) for k, p in (("a", pa), ("b", pb), ("c", pc))}
    ab = max(abs(x - z) for x, z in zip(la, lb))
    bc = max(abs(x - z) for x, z in zip(lb, lc))
    ab_late = sum(abs(x - z) for x, z in zip(la[-20:], lb[-20:])) / 20
    bc_late = sum(abs(x - z) for x, z in zip(lb[-20:], lc[-20:])) / 20
    with capsys.disabled():
        print(f"\n[model equivalence] {STEPS} steps of batch {BATCH}, held-out {HELD_OUT} molecules ({float(y.mean()):.2f} positive); "
              f"oracle {t_oracle:.0f} s on the CPU")
        print("  run                                   loss[0]  loss[39]  loss[79]   AUC    logAUC")
        for k, nm, ls in (("a", "(a) reference path (oracle, own argmax)", la), ("b", "(b) build", lb), ("c", "(c) build, second seed", lc)):
            print(f"  {nm:38s} {ls[0]:7.4f}  {ls[39]:7.4f}  {ls[79]:7.4f}  {met[k][0]:6.3f}  {met[k][1]:6.3f}")
        print(f"  max |loss_a - loss_b| = {ab:.2e} (mean of the last 20: {ab_late:.2e});  max |loss_b - loss_c| = {bc:.2e} ({bc_late:.2e})")
    # the task is learnt by all three
    for ls in (la, lb, lc):
        assert sum(ls[-10:]) / 10 < 0.85 * (sum(ls[:10]) / 10)
    # (b) follows (a) step for step, inside what a second seed does (measured: a sixth of it at the worst step, a quarter to a
    # third over the last 20 steps, where two trajectories that differ in the last bit have had 60 steps to drift apart; the
    # oracle's own run-to-run differences on the CPU -- thread-count dependent sums -- are part of that)
    assert ab <= 0.5 * bc, (ab, bc)
    assert ab_late <= 0.5 * bc_late + 2e-3, (ab_late, bc_late)
    # held-out metrics of (a) and (b) within the (b)-(c) spread (floors: the metrics' own resolution on 384 molecules)
    d_auc, d_log = abs(met["a"][0] - met["b"][0]), abs(met["a"][1] - met["b"][1])
    assert d_auc <= abs(met["b"][0] - met["c"][0]) + 0.01, (met, d_auc)
    assert d_log <= abs(met["b"][1] - met["c"][1]) + 0.02, (met, d_log)
