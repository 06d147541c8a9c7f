"""Packed molecule shards (molkgnn_amd/shards.py): format round trip, the loader's batch sequence and its
rank split on the CPU; on the GPU the loader's batches through the HIP builders and the model against the directly
collated batch."""
import numpy as np
import pytest
import torch

from molkgnn_amd import shards as S
from molkgnn_amd.synthetic import make_batch


def _same(a, b, fields=("x", "p", "edge_index", "edge_attr", "batch", "y", "assay_id")):
    for f in fields:
        ta, tb = getattr(a, f).cpu(), getattr(b, f).cpu()
        assert ta.shape == tb.shape and torch.equal(ta.to(tb.dtype), tb), f


def _slice(b, m0, m1):
    """Molecules [m0, m1) of a collated batch, re-collated by hand (the definition the loader must match)."""
    atoms = ((b.batch >= m0) & (b.batch < m1)).nonzero().view(-1)
    a0 = int(atoms[0])
    em = b.batch[b.edge_index[0]]
    edges = ((em >= m0) & (em < m1)).nonzero().view(-1)
    from molkgnn_amd.receptive_field import GraphBatch
    return GraphBatch(x=b.x[atoms], p=b.p[atoms], edge_index=b.edge_index[:, edges] - a0, edge_attr=b.edge_attr[edges],
                      batch=b.batch[atoms] - m0, y=b.y[m0:m1], assay_id=b.assay_id[m0:m1])


def test_shard_round_trip_and_loader_sequence(tmp_path):
    whole = make_batch(300, seed=4, assay="all9", with_receptive_fields=False)
    path = str(tmp_path / "a.mkgs")
    S.write_shard(path, whole)
    sh = S.Shard(path)
    assert (sh.n_molecules, sh.n_atoms, sh.n_edges) == (300, whole.x.shape[0], whole.edge_index.shape[1])
    assert sh.x_dim == 28 and sh.e_dim == 7 and sh.p_dim == 3
    assert np.array_equal(sh.x, whole.x.numpy()) and np.array_equal(sh.edge_attr, whole.edge_attr.numpy())
    _same(S.collate(sh, 0, 300, "cpu"), whole)
    loader = S.ShardLoader([path], batch_size=64, device="cpu")
    got = list(loader)
    assert [b.num_graphs for b in got] == [64, 64, 64, 64, 44] and len(loader) == 5
    for k, b in enumerate(got):
        _same(b, _slice(whole, 64 * k, min(300, 64 * k + 64)))
        assert int(b.mol_ptr[-1]) == b.x.shape[0] and int(b.mol_ptr[0]) == 0
    assert [b.num_graphs for b in S.ShardLoader([path], 64, drop_last=True)] == [64] * 4


def test_rank_split_is_a_partition_of_the_single_rank_stream(tmp_path):
    paths = S.write_shards(str(tmp_path), [make_batch(n, seed=10 + i, with_receptive_fields=False) for i, n in enumerate((100, 37, 64))])
    one = S.ShardLoader(paths, 32).plan()
    parts = [S.ShardLoader(paths, 32, rank=r, world=3).plan() for r in range(3)]
    assert sorted(sum(parts, [])) == sorted(one) and all(parts[r] == one[r::3] for r in range(3))
    assert sum(m1 - m0 for _, m0, m1 in one) == 201
    with pytest.raises(ValueError):
        S.ShardLoader(paths, 32, rank=3, world=3)


def test_reader_rejects_foreign_and_truncated_files(tmp_path):
    bad = tmp_path / "bad.mkgs"
    bad.write_bytes(b"\0" * 300)
    with pytest.raises(ValueError, match="not a molecule shard"):
        S.Shard(str(bad))
    good = str(tmp_path / "g.mkgs")
    S.write_shard(good, make_batch(20, seed=1, with_receptive_fields=False))
    data = open(good, "rb").read()
    (tmp_path / "cut.mkgs").write_bytes(data[:len(data) - 4096])
    with pytest.raises(ValueError, match="does not fit"):
        S.Shard(str(tmp_path / "cut.mkgs"))
    unsorted = make_batch(20, seed=1, with_receptive_fields=False)
    unsorted.batch = unsorted.batch.flip(0)
    with pytest.raises(ValueError, match="sorted"):
        S.write_shard(str(tmp_path / "u.mkgs"), unsorted)


@pytest.mark.gpu
def test_loader_batches_on_the_gpu_feed_the_model_like_direct_batches(tmp_path):
    from molkgnn_amd.receptive_field import attach_receptive_fields
    from molkgnn_amd.train import GNNModel
    dev = torch.device("cuda:0")
    whole = make_batch(700, seed=21, assay="all9", with_receptive_fields=False)
    paths = S.write_shards(str(tmp_path), [whole])
    torch.manual_seed(0)
    model = GNNModel(num_layers=3, ffn_dropout_rate=0.0).to(dev).eval()
    n = 0
    for k, b in enumerate(S.ShardLoader(paths, 256, device=dev, prefetch=2)):
        ref = _slice(whole, 256 * k, min(700, 256 * k + 256))
        _same(b, ref)
        assert b.x.is_cuda
        with torch.no_grad():
            got = model(attach_receptive_fields(b))[0]
            want = model(attach_receptive_fields(ref.to(dev)))[0]
        assert torch.equal(got, want)
        n += b.num_graphs
    assert n == 700


def test_fixed_shape_loader_pads_on_the_host_exactly_like_pad_batch(tmp_path):
    """ShardLoader(fixed_shape=True): the batch padded while it is staged, directly in StaticBatch's flat layout, must be
    byte for byte what padding.pack(padding.pad_batch(..)) makes of the collated batch; the shape is the common shape of
    the rank's batches, from the shard's degree prefix sums."""
    from molkgnn_amd import padding as P
    whole = make_batch(200, seed=6, assay="all9", with_receptive_fields=False)
    paths = S.write_shards(str(tmp_path), [whole])
    sh = S.Shard(paths[0])
    assert sh.degree_histogram(0, 200) == P.degree_histogram(whole)
    assert sh.degree_histogram(17, 90) == P.degree_histogram(_slice(whole, 17, 90))
    loader = S.ShardLoader(paths, 64, device="cpu", fixed_shape=True)
    plan = loader.plan()
    assert [m1 - m0 for _, m0, m1 in plan] == [64, 64, 64]                      # the short tail batch is dropped
    want_shape = P.fixed_shape([P.degree_histogram(_slice(whole, m0, m1)) for _, m0, m1 in plan])
    assert loader.shape == want_shape
    got = list(loader)
    assert len(got) == 3
    for (si, m0, m1), pb in zip(plan, got):
        ref = P.pack(P.pad_batch(_slice(whole, m0, m1), want_shape, 64))
        assert pb.flat.numel() == ref.flat.numel()
        g = pb.unpack((28, 3, 7))
        for k in P.StaticBatch.FIELDS:
            assert torch.equal(getattr(g, k), getattr(ref, k)), k
        assert pb.bucket_sizes == list(ref.bucket_sizes) and pb.n_valid_molecules == 64 and pb.num_graphs == ref.num_graphs
        sb = P.StaticBatch(ref)
        sb.load(pb)                                        # a PackedBatch loads like a packed GraphBatch
        assert torch.equal(sb.flat, pb.flat)


def _expand_on_host(cb, dims):
    """What mkgnn_expand_batch makes of a CompactBatch, in numpy (the definition the kernel must match)."""
    table, _ = S.compact_layout(cb.shape, cb.n_valid_molecules, *dims)
    host = cb.flat.cpu().numpy()
    f = {k: host[off:off + nbytes].view(dt).reshape(shp) for k, off, shp, dt, nbytes in table}
    ij = f["bond_ij"].astype(np.int64)
    ei = np.empty((2, 2 * ij.shape[0]), dtype=np.int64)
    ei[0, 0::2], ei[1, 0::2], ei[0, 1::2], ei[1, 1::2] = ij[:, 0], ij[:, 1], ij[:, 1], ij[:, 0]
    ea = np.repeat(f["bond_attr"].astype(np.float32), 2, axis=0)
    mp = f["mol_ptr"].astype(np.int64)
    bt = np.repeat(np.arange(mp.shape[0] - 1, dtype=np.int64), np.diff(mp))
    return {"x": f["x"], "p": f["p"], "edge_index": ei, "edge_attr": ea, "batch": bt, "y": f["y"], "mol_ptr": f["mol_ptr"],
            "atom_mol": bt.astype(np.int32), "n_valid_atoms": f["n_valid_atoms"]}


def test_compact_wire_form_expands_to_the_padded_batch(tmp_path):
    """ShardLoader(fixed_shape=True, compact=True): every bond once as an int32 pair with byte-valued attributes, nothing
    derived -- expanded, exactly the fields of the full fixed-shape form; 40 % fewer bytes; refused for shards whose
    bonds are not reversed pairs with shared byte-valued attributes."""
    whole = make_batch(200, seed=6, assay="all9", with_receptive_fields=False)
    paths = S.write_shards(str(tmp_path), [whole])
    assert S.Shard(paths[0]).compact_ok
    full = list(S.ShardLoader(paths, 64, device="cpu", fixed_shape=True))
    comp = list(S.ShardLoader(paths, 64, device="cpu", fixed_shape=True, compact=True))
    assert len(full) == len(comp) == 3
    for pb, cb in zip(full, comp):
        assert cb.flat.numel() < 0.65 * pb.flat.numel()
        want = pb.unpack((28, 3, 7))
        got = _expand_on_host(cb, (28, 3, 7))
        for k in ("x", "p", "edge_index", "edge_attr", "batch", "y", "mol_ptr", "atom_mol", "n_valid_atoms"):
            assert np.array_equal(got[k], getattr(want, k).numpy()), k
    odd = make_batch(50, seed=3, with_receptive_fields=False)
    odd.edge_attr = odd.edge_attr + 0.5                     # not byte-valued
    S.write_shard(str(tmp_path / "odd.mkgs"), odd)
    assert not S.Shard(str(tmp_path / "odd.mkgs")).compact_ok
    with pytest.raises(ValueError, match="compact"):
        S.ShardLoader([str(tmp_path / "odd.mkgs")], 16, fixed_shape=True, compact=True)


@pytest.mark.gpu
def test_compact_batches_expand_on_the_gpu_into_the_static_buffers(tmp_path):
    """padding.CompactStaticBatch: load + mkgnn_expand_batch give the tensors of the full fixed-shape batch, bit for bit."""
    from molkgnn_amd import padding as P
    dev = torch.device("cuda:0")
    whole = make_batch(300, seed=12, assay="all9", with_receptive_fields=False)
    paths = S.write_shards(str(tmp_path), [whole])
    full = list(S.ShardLoader(paths, 128, device=dev, fixed_shape=True))
    loader = S.ShardLoader(paths, 128, device=dev, fixed_shape=True, compact=True)
    csb = P.CompactStaticBatch(loader.shape, 128, 28, 3, 7, dev)
    for pb, cb in zip(full, loader):
        csb.load(cb)
        csb.expand()
        torch.cuda.synchronize()
        want = pb.unpack((28, 3, 7))
        for k in P.StaticBatch.FIELDS:
            assert torch.equal(getattr(csb.data, k), getattr(want, k)), k
        assert csb.data.bucket_sizes == want.bucket_sizes and csb.data.num_graphs == want.num_graphs


@pytest.mark.gpu
def test_training_from_shards_through_one_graph_learns():
    """tools/train_from_shards.py, shortened: shards -> fixed-shape compact loader -> CompactStaticBatch -> one captured
    graph (expand, builders, forward, backward with deferred bank gradients, AdamW) replayed over three epochs.  The label
    is a property of the graph, so the loss must fall and the held-out AUC must rise."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("train_from_shards", os.path.join(os.path.dirname(os.path.dirname(__file__)),
                                                                                   "tools", "train_from_shards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    losses, before, after = mod.run(molecules_per_shard=2048, n_shards=4, batch_size=512, epochs=4, log=lambda *_: None)
    assert losses[-1] < 0.6 * losses[0] and after[1] > max(0.8, before[1] + 0.2), (losses, before, after)
