"""Data-parallel plumbing on CPU: two gloo ranks, flat-buffer gradient all-reduce, molecule sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from molkgnn_amd import dp
    from molkgnn_amd.train import GNNModel, configure_optimizer
    assert dp.init_process_group_from_env("gloo") == world
    torch.manual_seed(7)                                   # identical replicas
    model = GNNModel(num_layers=2, kernels_1hop=(2, 2, 2, 2), kernels_Nhop=(2, 2, 2, 2), hidden_dim=8, ffn_hidden_dim=8)
    names = [n for n, _ in model.named_parameters()]
    reducer = dp.FlatGradAllReduce(model.parameters(), dp.NEVER_TRAINED, names)
    # the flat buffer covers exactly the parameters that can receive a gradient
    covered = {id(p) for p in reducer.params}
    for n, p in model.named_parameters():
        never = any(n.startswith(s[1:]) if s.startswith("^") else s in n for s in dp.NEVER_TRAINED)
        assert (id(p) in covered) == (not never), n
    # rank-dependent gradients; rank 1 "has no degree-4 atoms": those gradients stay None there
    g = torch.Generator().manual_seed(100 + rank)
    local = {}
    for n, p in model.named_parameters():
        if id(p) not in covered:
            continue
        if rank == 1 and "trainable_kernelconv_set.3." in n:
            continue
        if "layers.1.trainable_kernelconv_set.2." in n:       # "no degree-3 atom on ANY rank" in layer 1
            continue
        p.grad = torch.randn(p.shape, generator=g)
        local[n] = p.grad.clone()
    reducer.reduce()
    torch.save({"local": local, "reduced": {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    # one optimiser step keeps the replicas identical
    opt = configure_optimizer(model, weight_decay=1e-3, lr=1e-2, fused=False)
    assert len(opt.param_groups) == 2 and opt.param_groups[0]["weight_decay"] == 0
    nodecay = {id(p) for p in opt.param_groups[0]["params"]}
    for n, p in model.named_parameters():
        is_kernel = ("x_center" in n) or ("p_support" in n) or ("x_support" in n) or \
                    ("edge_attr_support" in n and "edge_attr_support_sc" not in n)
        assert (id(p) in nodecay) == is_kernel, n
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    assert list(dp.shard_indices(7, rank, world)) == list(range(rank, 7, world))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert set(r0["reduced"]) == set(r1["reduced"]) and len(r0["reduced"]) > 20
    # a bank without a gradient on every rank keeps none (the 1-GPU optimiser skips it: no step count, no moment decay);
    # a bank without a gradient on ONE rank receives the other rank's (halved) there too
    assert not any("layers.1.trainable_kernelconv_set.2." in n for n in r0["reduced"])
    assert any("trainable_kernelconv_set.3." in n for n in r1["reduced"])
    for n, g0 in r0["reduced"].items():
        assert torch.equal(g0, r1["reduced"][n]), n                      # same result on both ranks
        want = (r0["local"][n] + r1["local"].get(n, torch.zeros_like(g0))) / 2
        assert torch.allclose(g0, want, atol=1e-7), n                    # mean over ranks, missing = zero


def test_single_process_is_a_no_op():
    from molkgnn_amd import dp
    lin = torch.nn.Linear(3, 2)
    lin.weight.grad = torch.ones_like(lin.weight)
    red = dp.FlatGradAllReduce(lin.parameters())
    red.reduce()
    assert torch.equal(lin.weight.grad, torch.ones_like(lin.weight)) and lin.bias.grad is None
    assert red.nbytes == 4 * (6 + 2) + 4 * 2              # gradient slots + one has-gradient flag per parameter


def test_reducer_covers_the_trained_readout_weights():
    """gnn_model.graph_embedding_lin1 / lin2 are trained (MolKGNNNet.py:144-146) and must be in the all-reduce; only
    GNNModel's own lin1 / lin2 (model.py:147-148) are unused.  (A substring match on "lin1" once dropped the former.)"""
    from molkgnn_amd import dp
    from molkgnn_amd.train import GNNModel
    model = GNNModel(num_layers=1, kernels_1hop=(1, 1, 1, 1), kernels_Nhop=(1, 1, 1, 1), hidden_dim=4, ffn_hidden_dim=4)
    names = [n for n, _ in model.named_parameters()]
    reducer = dp.FlatGradAllReduce(model.parameters(), dp.NEVER_TRAINED, names)
    covered = {n for n, p in model.named_parameters() if any(p is q for q in reducer.params)}
    for n in ("gnn_model.graph_embedding_lin1.weight", "gnn_model.graph_embedding_lin2.bias", "ffn.weight",
              "gnn_model.node_batch_norm.weight", "gnn_model.gnn.layers.0.trainable_kernelconv_set.1.x_support"):
        assert n in covered, n
    for n in ("lin1.weight", "lin2.bias", "gnn_model.graph_embedding_linear.weight", "gnn_model.edge_batch_norm.weight"):
        assert n in names and n not in covered, n


def _worker_static_grads(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from molkgnn_amd import dp
    assert dp.init_process_group_from_env("gloo") == world
    torch.manual_seed(3)
    lin = torch.nn.Linear(4, 3)
    red = dp.FlatGradAllReduce(lin.parameters())
    # two captured graphs = two sets of gradient tensors; p.grad names only the last one's
    g = torch.Generator().manual_seed(10 + rank)
    sets = []
    for _ in range(2):
        for p in lin.parameters():
            p.grad = torch.randn(p.shape, generator=g)
        sets.append(red.grads())
    before = [[t.clone() for t in s] for s in sets]
    red.reduce(sets[0])                                    # "replay of graph 0": its tensors are averaged ...
    after = [[t.clone() for t in s] for s in sets]
    red.sum_into_flat(sets[1])                             # the flat-buffer form: a sum, nothing copied back
    saved = {"before": before, "after": after, "flat": [v.clone() for v in red.views],
             "set1_after_sum": [t.clone() for t in sets[1]], "flags_full": red.flags.clone()}
    # a gradient list with a hole (this rank's graph has no tensor for the bias on rank 1 only)
    holed = list(sets[1])
    if rank == 1:
        holed[1] = None
    red.sum_into_flat(holed)
    saved["flags_holed"] = red.flags.clone()
    saved["flat_holed"] = [v.clone() for v in red.views]
    try:
        red.reduce([sets[0][0], None])
        saved["raised"] = False
    except ValueError:
        saved["raised"] = True
    torch.save(saved, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_works_on_the_gradients_of_the_replayed_graph(tmp_path):
    """One hipGraph per resident batch: every graph writes into its own gradient tensors, and ``p.grad`` names the last
    capture's.  ``reduce(grads)`` must average exactly the given set and leave the other alone (bench.py passes each
    graph's list; reducing ``p.grad`` instead left the replicas out of sync -- caught by bench.py's replica check)."""
    port = _free_port()
    mp.spawn(_worker_static_grads, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    for k in range(2):                                     # weight, bias
        want = (r0["before"][0][k] + r1["before"][0][k]) / 2
        assert torch.allclose(r0["after"][0][k], want, atol=1e-7) and torch.equal(r0["after"][0][k], r1["after"][0][k])
        assert torch.equal(r0["after"][1][k], r0["before"][1][k]) and torch.equal(r1["after"][1][k], r1["before"][1][k])
        # sum_into_flat: the views hold the SUM over ranks of the given set, which itself is untouched
        assert torch.allclose(r0["flat"][k], r0["before"][1][k] + r1["before"][1][k], atol=1e-6)
        assert torch.equal(r0["flat"][k], r1["flat"][k]) and torch.equal(r0["set1_after_sum"][k], r0["before"][1][k])
    # the has-gradient flags ride in the same all-reduce: number of ranks with a gradient, identical on both ranks
    for r in (r0, r1):
        assert r["flags_full"].tolist() == [2.0, 2.0] and r["flags_holed"].tolist() == [2.0, 1.0]
        assert r["raised"]                                 # reduce(grads) refuses a None entry instead of skipping it
    # rank 1 had no bias gradient: both ranks still see rank 0's in the flat view (and would apply the same update)
    assert torch.allclose(r0["flat_holed"][1], r0["before"][1][1], atol=1e-6)
    assert torch.equal(r0["flat_holed"][1], r1["flat_holed"][1])


def _buffer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from molkgnn_amd import dp
    assert dp.init_process_group_from_env("gloo") == world
    torch.manual_seed(3)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 4), torch.nn.BatchNorm1d(4))
    net.train()
    g = torch.Generator().manual_seed(20 + rank)           # every rank its own batches: its own running statistics
    steps = 3 + rank                                       # (and, here, its own step count)
    for _ in range(steps):
        net(torch.randn(16, 5, generator=g) * (1 + rank) + rank)
    mine = {n: b.clone() for n, b in net.named_buffers()}
    sync = dp.BufferSync(net)
    before = sync.max_abs_diff()
    gathered = {}
    for n, b in mine.items():
        if b.dtype.is_floating_point:
            t = [torch.zeros_like(b) for _ in range(world)]
            dist.all_gather(t, b)
            gathered[n] = torch.stack(t).mean(dim=0)
    sync.average()
    after = sync.max_abs_diff()
    ok_mean = all(torch.allclose(b, gathered[n], atol=1e-6) for n, b in net.named_buffers() if b.dtype.is_floating_point)
    counts = [int(b) for n, b in net.named_buffers() if not b.dtype.is_floating_point]
    # DDP's form: everything from rank 0
    net2 = torch.nn.BatchNorm1d(3)
    net2.train()
    net2(torch.randn(8, 3, generator=g) + rank)
    s2 = dp.BufferSync(net2)
    d2 = s2.max_abs_diff()
    s2.broadcast()
    torch.save({"before": before, "after": after, "ok_mean": ok_mean, "counts": counts, "d2_before": d2, "d2_after": s2.max_abs_diff(),
                "rm2": net2.running_mean.clone()}, os.path.join(out_dir, f"buf{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_buffer_sync_averages_running_statistics_over_the_ranks(tmp_path):
    """dp.BufferSync: BatchNorm's running statistics are per-rank state that the gradient
    all-reduce never touches; ``average()`` leaves every rank with the mean over ranks (integer buffers from rank 0),
    ``broadcast()`` with rank 0's, and ``max_abs_diff()`` -- bench.py's dp_buffers.max_abs_diff -- is 0 afterwards."""
    port = _free_port()
    mp.spawn(_buffer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "buf0.pt"), torch.load(tmp_path / "buf1.pt")
    for r in (r0, r1):
        assert r["before"] > 1e-3 and r["after"] == 0.0 and r["ok_mean"]
        assert r["counts"] == [3, 3]                        # rank 0's step count on both ranks
        assert r["d2_before"] > 1e-3 and r["d2_after"] == 0.0
    assert torch.equal(r0["rm2"], r1["rm2"])
