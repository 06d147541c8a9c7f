"""CPU-only checks: the C-ABI library loads and exports what include/*.h declares,
the host-side index logic, and the synthetic generator.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(REPO, "include", "molkgnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mkgnn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from molkgnn_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `make -C molkgnn_amd/csrc` (or __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 7
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib.EXPORTS)
    lib.mkgnn_abi_version.restype = ctypes.c_int
    assert lib.mkgnn_abi_version() == _lib.ABI_VERSION
    # host-only entry point: workspace sizing needs no GPU
    typed = _lib.load()
    small = typed.mkgnn_workspace_bytes(_lib.Int32x4(10, 20, 30, 50), 28, 7, 100, 220)
    big = typed.mkgnn_workspace_bytes(_lib.Int32x4(10, 20, 30, 50), 110, 7, 100000, 220000)
    assert 0 < small < big


def test_no_cpu_fallback():
    from molkgnn_amd._lib import MolKGNNLibraryError
    from molkgnn_amd.kernels import KernelSetConv
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(2, seed=1)
    ksc = KernelSetConv(2, 2, 2, 2, D=3, node_attr_dim=28, edge_attr_dim=7)
    with pytest.raises(MolKGNNLibraryError, match="no CPU fallback"):
        ksc(is_last_layer=False, data=b, save_score=False)


def test_receptive_fields_follow_the_reference_layout():
    """Brute-force restatement of wrapper.py:567-635 on a small batch."""
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(6, seed=3)
    ei, ea, p = b.edge_index, b.edge_attr, b.p
    n = b.x.shape[0]
    deg = torch.bincount(ei[0], minlength=n)
    assert int(deg.min()) >= 1 and int(deg.max()) <= 4
    # bonds are stored as consecutive (i, j), (j, i) with identical attributes (wrapper.py:152-156)
    assert torch.equal(ei[:, 0::2], ei[:, 1::2].flip(0)) and torch.equal(ea[0::2], ea[1::2])
    for d in range(1, 5):
        sel = getattr(b, f"selected_index_deg{d}")
        assert torch.equal(sel, (deg == d).nonzero(as_tuple=True)[0])
        nei, nei_p, nei_e = [], [], []
        for c in sel.tolist():
            e = (ei[0] == c).nonzero()[:, 0]                     # get_neighbor_index
            nei.append(ei[1, e])
            nei_p.append(p[ei[1, e]])
            nei_e.append(ea[2 * (e // 2)])                       # get_edge_attr_support_from_center_node
        if sel.numel():
            assert torch.equal(getattr(b, f"nei_index_deg{d}"), torch.stack(nei).reshape(-1))
            assert torch.equal(getattr(b, f"nei_p_deg{d}"), torch.stack(nei_p))
            assert torch.equal(getattr(b, f"nei_edge_attr_deg{d}"), torch.stack(nei_e))
            assert torch.equal(getattr(b, f"p_focal_deg{d}"), p[sel])
        else:
            assert getattr(b, f"nei_index_deg{d}").numel() == 0


def test_plan_scatter_and_csr():
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(7, seed=4)
    plan = plan_from_data(b)
    assert plan_from_data(b) is plan                      # cached on the batch
    n = b.x.shape[0]
    rowptr, rows = plan.scatter
    dest = []
    for bk in plan.buckets:
        dest.append(torch.cat([bk.sel.view(-1, 1), bk.nei.view(bk.count, bk.degree)], 1).reshape(-1))
    dest = torch.cat(dest)
    assert rows.numel() == dest.numel() == plan.n_focal + plan.n_slots
    for j in range(n):
        seg = rows[rowptr[j]:rowptr[j + 1]].long()
        assert torch.all(dest[seg] == j)
        assert torch.all(seg[1:] > seg[:-1])              # fixed order -> reproducible sums
    # every atom is focal once and a neighbour deg(j) times
    deg = torch.bincount(b.edge_index[0], minlength=n)
    assert torch.equal((rowptr[1:] - rowptr[:-1]).long(), deg + 1)
    rp, col = plan.csr_in
    v = torch.randn(n, 3)
    ref = torch.zeros_like(v).index_add_(0, b.edge_index[1], v[b.edge_index[0]])
    got = torch.stack([v[col[rp[i]:rp[i + 1]].long()].sum(0) for i in range(n)])
    assert torch.allclose(got, ref, atol=1e-6)


def test_synthetic_molecules_have_the_surveyed_shape():
    from molkgnn_amd.synthetic import ASSAY_SIZES, degree_histogram, make_batch
    assert ASSAY_SIZES["1798"] == 61832 and ASSAY_SIZES["435008"] == 218156
    b = make_batch(2000, seed=1798)
    n = b.x.shape[0]
    assert 24.0 < n / 2000 < 26.0
    assert 50.0 < b.edge_index.shape[1] / 2000 < 56.0
    h = np.array(degree_histogram(b)[:5]) / n
    assert h[0] == 0 and abs(h[1] - 0.22) < 0.05 and abs(h[2] - 0.44) < 0.05 and abs(h[3] - 0.30) < 0.05 and h[4] < 0.08
    assert b.x.shape[1] == 28 and b.edge_attr.shape[1] == 7 and b.p.shape[1] == 3
    b2 = make_batch(2000, seed=1798)
    assert torch.equal(b.x, b2.x) and torch.equal(b.edge_index, b2.edge_index)
    dup = make_batch(50, seed=5, duplicate_fraction=0.5)
    nei = dup.nei_index_deg2.view(-1, 2)
    assert (dup.x[nei[:, 0]] == dup.x[nei[:, 1]]).all(dim=1).any()


def test_module_surface_matches_the_reference_names():
    from molkgnn_amd.KernelLayer import MolGCN
    from molkgnn_amd.kernels import KernelConv
    g = MolGCN(num_layers=2, num_kernel1_1hop=1, num_kernel2_1hop=2, num_kernel3_1hop=3, num_kernel4_1hop=4,
               num_kernel1_Nhop=2, num_kernel2_Nhop=2, num_kernel3_Nhop=2, num_kernel4_Nhop=2, x_dim=6, p_dim=3,
               edge_attr_dim=2)
    keys = list(g.state_dict().keys())
    assert keys[0] == "layers.0.trainable_kernelconv_set.0.x_center"
    per = ["x_center", "x_support", "edge_attr_support", "p_support", "length_sc_weight", "angle_sc_weight",
           "center_attr_sc_weight", "support_attr_sc_weight", "edge_attr_support_sc_weight"]
    assert [k.split(".")[-1] for k in keys[:9]] == per
    assert g.layers[1].trainable_kernelconv_set[2].x_support.shape == (2, 3, 10)   # N-hop width = 1+2+3+4
    assert g.num_kernels(0) == 10 and g.num_kernels(1) == 8
    with pytest.raises(Exception, match="positional"):
        g(1)
    with pytest.raises(Exception, match="at least one"):
        MolGCN(num_layers=0)
    k = KernelConv(L=4, D=3, num_supports=3, node_attr_dim=6, edge_attr_dim=2)
    assert k.get_num_kernels() == 4 and k.support_attr_sc_weight.dim() == 0


def test_molecule_segments_and_readout_refuses_cpu():
    from molkgnn_amd import readout as R
    from molkgnn_amd._lib import MolKGNNLibraryError
    batch = torch.tensor([0, 0, 1, 1, 1, 3])
    seg = R.MoleculeSegments(batch, 5)
    assert seg.sorted and seg.mol_ptr.tolist() == [0, 2, 5, 5, 6, 6] and seg.atom_mol.tolist() == batch.tolist()
    assert not R.MoleculeSegments(torch.tensor([1, 0, 1]), 2).sorted
    with pytest.raises(ValueError):
        R.MoleculeSegments(torch.tensor([0, 4]), 3)
    assert R.readout_supported(110, 32, 32) and not R.readout_supported(200, 32, 32)
    lin = torch.nn.Linear(4, 4)
    with pytest.raises(MolKGNNLibraryError):
        R.readout(torch.zeros(6, 4), lin, lin, None, batch, 5)
    with pytest.raises(MolKGNNLibraryError):
        R.batch_norm(torch.zeros(6, 4), torch.nn.BatchNorm1d(4))


def test_evaluation_metrics_match_reference_golden():
    """evaluation.py:11-127 (logAUC, AUC, PPV, accuracy, F1) against numbers produced by the reference's own
    functions (tests/golden/make_golden_metrics.py)."""
    import math
    import os
    from molkgnn_amd import evaluation as E
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_metrics.npz"))
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/y")})
    assert len(names) == 6
    for name in names:
        y, s = torch.from_numpy(g[f"{name}/y"]), torch.from_numpy(g[f"{name}/score"])
        for key, got in (("logauc", E.calculate_logAUC(y, s)), ("logauc_wide", E.calculate_logAUC(y, s, FPR_range=(0.01, 0.5))),
                         ("auc", E.calculate_auc(y, s)), ("ppv", E.calculate_ppv(y, s)), ("ppv_cut", E.calculate_ppv(y, s, cutoff=0.8)),
                         ("accuracy", E.calculate_accuracy(y, s)), ("f1", E.calculate_f1_score(y, s))):
            want = float(g[f"{name}/{key}"])
            assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (name, key, got, want)
    # one class only: the reference's `except: res = -1` (evaluation.py:81-86) fires under its pinned scikit-learn 1.0.2
    # (ValueError); the container's 1.7 -- which generated the golden file -- returns NaN without raising
    assert E.calculate_auc(torch.zeros(10), torch.randn(10)) == -1 and E.calculate_auc(torch.ones(4), torch.randn(4)) == -1
    assert math.isnan(float(g["oneclass/auc"]))
    with pytest.raises(Exception):
        E.calculate_logAUC(torch.tensor([0, 1]), torch.tensor([0.1, 0.9]), FPR_range=(0.1, 0.01))


def test_oversampling_sampler_matches_reference_construction():
    """data.py:136-166: same weights and the same index stream as the reference's per-sample construction."""
    from torch.utils.data import WeightedRandomSampler
    from molkgnn_amd.sampling import oversampling_sampler, oversampling_weights
    g = torch.Generator().manual_seed(3)
    y = (torch.rand(500, generator=g) < 0.05).long()
    n_active = int(y.sum()); n_inactive = y.numel() - n_active
    ref_w = torch.tensor([(1. / n_inactive) if int(v) == 0 else (1. / n_active) for v in y])      # the reference's loop
    assert torch.equal(oversampling_weights(y), ref_w)
    gen = torch.Generator(); gen.manual_seed(42)
    ref = list(WeightedRandomSampler(weights=ref_w, num_samples=len(ref_w), generator=gen))
    assert list(oversampling_sampler(y, 42)) == ref
    assert abs(y[torch.tensor(ref)].float().mean().item() - 0.5) < 0.1                              # classes come out balanced


def test_header_is_plain_c_and_cpp(tmp_path):
    """The drop-in boundary is a C ABI: include/molkgnn_hip.h must compile as C99 and as C++ with nothing but the
    standard headers (no HIP, no torch types)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "molkgnn_hip.h"\nint main(void) { return (int)sizeof(mkgnn_kernel_bank) == 0; }\n')
    inc = os.path.join(root, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], check=True)
    if shutil.which("g++") is not None:
        subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)], check=True)


def test_plan_cache_is_not_fooled_by_recycled_addresses():
    """The plan cache is keyed by tensor addresses, which the allocator recycles: an entry must only be returned for
    the very tensor objects it was built from (other objects at the same address -- here views -- rebuild)."""
    from molkgnn_amd.plan import plan_from_lists_cached
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(6, seed=2)
    def lists(f):
        return [[f(getattr(b, f"{nm}_deg{d}")) for d in range(1, 5)]
                for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index")]
    same = lambda t: t
    alias = lambda t: t.view_as(t)                   # same address, same length, another tensor object
    n = b.x.shape[0]
    p1 = plan_from_lists_cached(n, *lists(same), b.edge_index)
    assert plan_from_lists_cached(n, *lists(same), b.edge_index) is p1
    held = lists(alias)
    p2 = plan_from_lists_cached(n, *held, b.edge_index)
    assert p2 is not p1
    assert torch.equal(p2.scatter[0], p1.scatter[0]) and torch.equal(p2.scatter[1], p1.scatter[1])


def test_plan_block_row_tables():
    """Host side of the block-row propagate: the degree table built from the buckets (0 = in no bucket) and the
    propagate CSR whose column entries carry the source atom's degree in bits 28..30."""
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    b = make_batch(9, seed=6)
    plan = plan_from_data(b)
    n = b.x.shape[0]
    deg = torch.bincount(b.edge_index[0], minlength=n)
    expect = torch.where(deg <= 4, deg, torch.zeros_like(deg)).to(torch.int8)
    assert torch.equal(plan.deg8, expect)
    rowptr, col = plan.csr_in
    rp2, packed = plan.csr_in_packed
    assert rp2 is rowptr and packed.dtype == torch.int32
    assert torch.equal(packed & 0x0FFFFFFF, col)
    assert torch.equal((packed >> 28) & 7, plan.deg8[col.long()].to(torch.int32))
    assert plan.block_rows_ok(110) and not plan.block_rows_ok(300) and not plan.block_rows_ok(0)


def test_fused_adamw_host_side():
    """FusedAdamW without a GPU: argument checks, refusal of CPU parameters (no CPU path), and the packed state
    (exp_avg | exp_avg_sq | step as views of one buffer) built from a torch.optim.AdamW state_dict."""
    from molkgnn_amd._lib import MolKGNNLibraryError
    from molkgnn_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.randn(3, 5))
    with pytest.raises(ValueError):
        FusedAdamW([p], betas=(1.0, 0.9))
    with pytest.raises(ValueError):
        FusedAdamW([{"params": [torch.nn.Parameter(torch.zeros(1))]} for _ in range(5)])
    opt = FusedAdamW([p], lr=1e-3)
    p.grad = torch.randn(3, 5)
    with pytest.raises(MolKGNNLibraryError, match="no CPU fallback"):
        opt.step()
    ref = torch.optim.AdamW([p], lr=1e-3)
    ref.step()
    opt2 = FusedAdamW([p], lr=1e-3)
    opt2.load_state_dict(ref.state_dict())
    st = opt2.state[p]
    buf = st["_packed"]
    from molkgnn_amd import _lib
    assert buf.numel() == 2 * 15 + 3 + 1 == _lib.load().mkgnn_adamw_state_floats(15) and float(st["step"]) == 1.0
    assert all(FusedAdamW._state_floats(n) == _lib.load().mkgnn_adamw_state_floats(n) for n in (1, 1023, 1024, 1025, 22000))
    assert float(buf[2 * 15 + 3]) == 1.0                      # the update's per-block copy of the step count follows the loaded one
    assert st["exp_avg"].data_ptr() == buf.data_ptr() and st["exp_avg"].shape == p.shape
    assert torch.equal(st["exp_avg"], ref.state[p]["exp_avg"]) and torch.equal(st["exp_avg_sq"], ref.state[p]["exp_avg_sq"])
    assert "_packed" not in next(iter(opt2.state_dict()["state"].values()))


def test_receptive_field_builder_matches_the_reference_transform():
    """G10: ``ToXAndPAndEdgeAttrForDeg.__call__`` of the reference itself (wrapper.py:637-672, imported unmodified by
    tests/golden/make_golden_rf.py) run per molecule, then collated: the batch-level builder must reproduce all 20
    tensors exactly -- molecules with shuffled bond order, a hub atom of degree 7 (in no bucket), a two-atom molecule
    (degrees 2-4 absent); and, molecule by molecule, the reference's own per-molecule outputs."""
    import numpy as np
    from molkgnn_amd.receptive_field import build_receptive_fields
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_receptive_fields.npz"))
    t = lambda k: torch.from_numpy(z[k])
    names = [f"{nm}_deg{d}" for d in range(1, 5) for nm in ("p_focal", "nei_p", "nei_edge_attr", "selected_index", "nei_index")]
    got = build_receptive_fields(t("batch/x"), t("batch/p"), t("batch/edge_index"), t("batch/edge_attr"))
    deg = torch.bincount(t("batch/edge_index")[0], minlength=t("batch/x").shape[0])
    assert int(deg.max()) == 7                         # the hub is really there
    for k in names:
        want = t("batch/" + k)
        assert got[k].numel() == want.numel(), k
        assert torch.equal(got[k].reshape(want.shape), want), k
    for i in range(int(z["num_molecules"])):
        g = build_receptive_fields(t(f"mol{i}/in_x"), t(f"mol{i}/in_p"), t(f"mol{i}/in_edge_index"), t(f"mol{i}/in_edge_attr"))
        for k in names:
            want = t(f"mol{i}/{k}")
            assert g[k].numel() == want.numel(), (i, k)
            if want.numel():
                assert torch.equal(g[k].reshape(want.shape), want.to(g[k].dtype)), (i, k)


def test_torch_library_shim_loads_and_registers_its_operators():
    """libmolkgnn_torch.so (csrc/torch_ops.cpp, TORCH_LIBRARY(molkgnn)): built by build(), loads on a CPU-only box, agrees with
    the C ABI's version and exposes the three operators (they themselves need a GPU)."""
    import torch
    from molkgnn_amd import _lib
    ops = _lib.load_torch_ops()
    assert int(ops.abi_version()) == _lib.ABI_VERSION
    for name in ("kernelsetconv_forward", "kernelsetconv_backward", "backward_join"):
        assert hasattr(torch.ops.molkgnn, name)


def test_no_lds_dma_site_was_merged_by_the_compiler(tmp_path):
    """The LDS base of an LDS-DMA (M0) is wave-uniform by construction in the streamed kernels.  LLVM's code sinking once
    merged the DMA of a ``if (lane < 32)`` region with the unconditional one behind it into ONE instruction whose LDS base
    was a phi of two destinations, made uniform by ``v_readfirstlane`` -- half of the lanes' rows landed in the other site's
    record (round 3, KC = 1; csrc/kgnn_fwd_stream.hip, ``SITE``).  Every site now carries its own immediate offset, which
    cannot be phi'd; this test compiles the two kernels that issue such DMAs to ISA and checks that no M0 write is fed by a
    ``v_readfirstlane``."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is not available")
    csrc = os.path.join(REPO, "molkgnn_amd", "csrc")
    procs = []
    for name in ("kgnn_fwd_stream", "kgnn_bwd_stream"):
        out = str(tmp_path / f"{name}.s")
        procs.append((name, out, subprocess.Popen(
            [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
             os.path.join(csrc, name + ".hip"), "-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for name, out, pr in procs:
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, err.decode()[-2000:]
        text = open(out).read()
        n_dma = len(re.findall(r"global_load_lds_dword", text))
        assert n_dma > 100, (name, n_dma)                     # (the kernels are in there)
        merged = re.findall(r"v_readfirstlane_b32 (s\d+), v\d+\n(?:[^\n]*\n){0,2}?\s*s_mov_b32 m0, \1\b", text)
        assert not merged, (name, len(merged))


def test_native_collate_writes_the_numpy_collate_byte_for_byte(tmp_path):
    """mkgnn_collate_compact (host code of the library, one call per batch on a loader thread) against shards.collate_compact
    (numpy, the definition): the same staging buffer byte for byte on ragged molecule ranges of several shards, padding atoms
    / bonds / molecules included; a range that does not fit the shape is refused with the same complaint."""
    import numpy as np
    from molkgnn_amd import padding as P, shards as S
    from molkgnn_amd._lib import MolKGNNLibraryError
    from molkgnn_amd.synthetic import make_batch
    raws = [make_batch(150 + 30 * i, seed=70 + i, assay="all9", with_receptive_fields=False) for i in range(3)]
    shards = [S.Shard(p) for p in S.write_shards(str(tmp_path), raws)]
    for B in (1, 37, 64):
        work = [(si, m0, m0 + B) for si, sh in enumerate(shards) for m0 in range(0, sh.n_molecules - B + 1, B)]
        shape = P.fixed_shape([shards[si].degree_histogram(m0, m1) for si, m0, m1 in work])
        for si, m0, m1 in work:
            sh = shards[si]
            _, total = S.compact_layout(shape, m1 - m0, sh.x_dim, sh.p_dim, sh.e_dim)
            a, b = np.full(total, 0xAB, dtype=np.uint8), np.full(total, 0xAB, dtype=np.uint8)
            S.collate_compact(sh, m0, m1, shape, a)
            S.collate_compact_native(sh, m0, m1, shape, b)
            assert np.array_equal(a, b), (B, si, m0)
    small = dict(shape)
    small["n2"] = 0
    sh = shards[0]
    buf = np.zeros(1 << 22, dtype=np.uint8)
    with pytest.raises(ValueError):
        S.collate_compact(sh, 0, 64, small, buf)
    with pytest.raises(MolKGNNLibraryError, match="degree 2"):
        S.collate_compact_native(sh, 0, 64, small, buf)
    with pytest.raises(MolKGNNLibraryError, match="too small"):
        S.collate_compact_native(sh, 0, 64, shape, buf[:1024])
    del shards, sh


def test_molecule_chunk_table_and_abi_structs():
    """Host side of the molecule-resident step (molkgnn_amd.molecule): the greedy chunk table keeps molecules whole, stays
    within the atom cap and 16 molecules per chunk, gives an oversized molecule a chunk of its own; the ctypes structs have
    the header's sizes (a C compile of include/molkgnn_hip.h); the shape query needs no GPU."""
    import subprocess
    import tempfile
    from molkgnn_amd import _lib
    from molkgnn_amd import molecule as M
    sizes = [25, 8, 8, 8, 40, 3, 30, 2] + [1] * 40 + [60, 33]
    ptr, big = M.chunk_molecules(sizes, 32)
    assert ptr[0] == 0 and ptr[-1] == len(sizes) and all(b > a for a, b in zip(ptr, ptr[1:]))
    per_chunk = [sum(sizes[a:b]) for a, b in zip(ptr, ptr[1:])]
    for (a, b), atoms in zip(zip(ptr, ptr[1:]), per_chunk):
        assert b - a <= M.MAX_MOLS
        assert atoms <= 32 or b - a == 1            # over the cap only as a single oversized molecule
    assert big == max(per_chunk) == 60 and sum(per_chunk) == sum(sizes)
    assert M.chunk_molecules([], 32) == ([0, 0], 0) or M.chunk_molecules([], 32)[0][-1] == 0
    src = '#include "molkgnn_hip.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu %d\\n", sizeof(mkgnn_molecule_layer), ' \
          'sizeof(mkgnn_molecule_net), sizeof(mkgnn_molecule_batch), MKGNN_MOLECULE_MAX_ATOMS);return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "sz.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), os.path.join(d, "sz.c"), "-o", os.path.join(d, "sz")], check=True)
        got = subprocess.run([os.path.join(d, "sz")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in got] == [ctypes.sizeof(_lib.MoleculeLayer), ctypes.sizeof(_lib.MoleculeNet),
                                     ctypes.sizeof(_lib.MoleculeBatch), M.MAX_ATOMS]
    # the shape query is host code: the reference's 3-layer model qualifies, a 5-layer one or a 113-kernel layer does not
    lib = _lib.load()

    def net(layers, counts, x_dim=28, H=32, G=32):
        st = _lib.MoleculeNet()
        st.num_layers, st.E = layers, 7
        F = x_dim
        for li in range(min(layers, 4)):
            st.layer[li].F = F
            for d in range(4):
                st.layer[li].bank[d].num_kernels = counts[d]
            F = sum(counts)
        st.readout.F, st.readout.H, st.readout.G = F, H, G
        return st
    assert lib.mkgnn_molecule_supported(ctypes.byref(net(3, (10, 20, 30, 50))), 28) == 1
    assert lib.mkgnn_molecule_supported(ctypes.byref(net(4, (1, 1, 1, 1))), 28) == 1
    assert lib.mkgnn_molecule_supported(ctypes.byref(net(5, (10, 20, 30, 50))), 28) == 0
    assert lib.mkgnn_molecule_supported(ctypes.byref(net(3, (10, 20, 30, 53))), 28) == 0        # K = 113 > 112
    assert lib.mkgnn_molecule_supported(ctypes.byref(net(3, (10, 20, 30, 50), x_dim=40)), 40) == 0
    assert lib.mkgnn_molecule_supported(ctypes.byref(net(3, (10, 20, 30, 50), H=65)), 28) == 0
    assert lib.mkgnn_molecule_workspace_bytes(ctypes.byref(net(3, (10, 20, 30, 50))), 28, 400, 16) > 7_000_000


def test_split_fp16_scales_are_exact_powers_of_two_over_every_exponent():
    """The power-of-two scales of the split-fp16 products (csrc/kgnn_split.h), evaluated on the host by the
    very functions the kernels compile (``mkgnn_debug_split_scales``): for EVERY biased exponent and random mantissas --
    the scale is a power of two, the unscale is exactly its reciprocal times the documented 2^-EXTRA, the scaled magnitude lands
    in [2^TARGET, 2^(TARGET + 1)) wherever the scale is not clamped (exponents 24 .. 230), and the clamp itself is where the
    header says; the row scale of the forward / of pre-split rows times its companion gives back 1 / |x| exactly."""
    import struct
    from hypothesis import given, settings, strategies as st
    from molkgnn_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.mkgnn_debug_split_scales.restype = ctypes.c_int
    lib.mkgnn_debug_split_scales.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32 * 6)]

    def f32(bits):
        return struct.unpack("<f", struct.pack("<I", bits))[0]

    def check(eb, mant):
        bits = (eb << 23) | mant
        out = (ctypes.c_uint32 * 6)()
        assert lib.mkgnn_debug_split_scales(bits, ctypes.byref(out)) == 0
        s_rows, u_rows, s_bank, u_bank, s_row, i_row = [int(v) for v in out]
        # (the row scale is 2^(exponent + 8) of 1 / max(|x|, 1e-8) <= 1e8: exponents above 127 + 27 do not occur; checked to 246)
        for s in (s_rows, u_rows, s_bank, u_bank) + ((s_row,) if eb < 255 - 8 else ()):
            assert s & 0x007FFFFF == 0 and 0 < (s >> 23) < 255, (eb, hex(s))          # a normal power of two
        # rows kernel: TARGET = 10, unscale carries 2^-12
        f = min(230, max(24, 254 + 10 - eb))
        assert s_rows >> 23 == f and (u_rows >> 23) == 254 - 12 - f
        assert np.float32(f32(s_rows)) * np.float32(f32(u_rows)) == np.float32(2.0 ** -12)
        if 24 <= 254 + 10 - eb <= 230 and 0 < eb < 255:
            scaled = np.float64(f32(bits)) * np.float64(f32(s_rows))
            assert 2.0 ** 10 <= scaled < 2.0 ** 11, (eb, scaled)
        # bank kernel: TARGET = 18 from the exponent, unscale = 1 / scale
        f = min(230, max(24, 254 + 18 - eb))
        assert s_bank >> 23 == f and np.float32(f32(s_bank)) * np.float32(f32(u_bank)) == np.float32(1.0)
        # row scale (amax read as 1 / |x|): 2^(exponent + 8) and the mantissa at exponent -8: their product is the value itself
        if 0 < eb < 255 - 8:
            assert s_row >> 23 == eb + 8 and (i_row & 0x007FFFFF) == mant and (i_row >> 23) == 127 - 8
            assert np.float32(f32(s_row)) * np.float32(f32(i_row)) == np.float32(f32(bits))

    for eb in range(0, 255):                               # every exponent, both ends of the mantissa range
        for mant in (0, 1, 0x400000, 0x7FFFFF):
            check(eb, mant)

    @settings(max_examples=300, deadline=None)
    @given(st.integers(min_value=0, max_value=254), st.integers(min_value=0, max_value=0x7FFFFF))
    def prop(eb, mant):
        check(eb, mant)
    prop()
