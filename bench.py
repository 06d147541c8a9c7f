"""Throughput of the 3-layer MolKGNN training step on AID-1798-shaped synthetic molecules.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  ``value`` = molecules/s of forward + backward (+ gradient all-reduce for
N > 1, + AdamW step) over all ranks, batches already resident in HBM.  ``roofline`` prices the
kernel-convolution forward of one N-hop layer (F = 110, the dominant kernels) against the HBM
roofline with the algorithmic byte count of SURVEY.md 8(d); ``cpu_baseline`` times the
reference-faithful CPU restatement (oracle/) on this box's host cores on a bounded sample.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TFLOPS = 157.3


def loss_backward(model, batch):
    """``loss -> backward`` as one unit (readout.deferred_tail_reduce: the fused tail's last reduction rides on the backward's helper
    stream); the loss is complete in stream order when this returns."""
    from molkgnn_amd.readout import deferred_tail_reduce
    from molkgnn_amd.train import backward as train_backward
    with deferred_tail_reduce(batch.x.device):
        loss = model.loss(batch)
        train_backward(loss)
    return loss


def products_mode(variant: str) -> dict:
    """What the streamed kernels multiply with (read from the same environment switches the library reads)."""
    fwd = "bf16 operands (similarity variant)" if variant == "bf16" else (
        "fp32 out of split fp16: 3 x v_mfma_f32_16x16x16_f16 on exact hi + lo operands, fp32 accumulation"
        if os.environ.get("MKGNN_FWD_SPLIT", "1") != "0" else "v_mfma_f32_16x16x4_f32")
    bwd = ("fp32 out of split fp16 (as the forward)" if os.environ.get("MKGNN_BWD_SPLIT", "1") != "0" else "v_mfma_f32_16x16x4_f32")
    split_rows = (variant in ("auto", "mfma") and os.environ.get("MKGNN_ROWS_SPLIT", "1") != "0"
                  and os.environ.get("MKGNN_FWD_SPLIT", "1") != "0" and os.environ.get("MKGNN_BWD_SPLIT", "1") != "0")
    return {"forward": fwd, "backward": bwd,
            "rows": ("h between two layers is written by propagate as the fp16 hi | lo halves the matrix instructions take "
                     "(pre-split rows, DESIGN 4.1f): same bytes, the forward's scores bit for bit; MKGNN_ROWS_SPLIT=0 keeps fp32 rows"
                     if split_rows else "fp32"),
            "accuracy": "fp32-grade: tests/test_scale_parity.py::test_split_fp16_products_are_fp32_grade "
            "(against float64: no worse than the fp32 matrix instructions)", "switches": "MKGNN_FWD_SPLIT=0 / MKGNN_BWD_SPLIT=0 restore the fp32 instructions"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=4096, help="molecules per step per GPU")
    ap.add_argument("--assay", default="1798")
    ap.add_argument("--variant", default="auto", choices=["auto", "generic", "mfma", "bf16"])
    ap.add_argument("--distinct-batches", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+backward(+all-reduce) only")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying one hipGraph per resident batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--roofline-reps", type=int, default=20)
    ap.add_argument("--windows", type=int, default=0, help="timed windows of --steps steps each (0: as many as make >= 0.5 s of timed work, at least 5)")
    ap.add_argument("--dp-one-graph", action="store_true",
                    help="N > 1: capture the gradient all-reduce INTO the step's graph (one graph per batch, as on one GPU) instead "
                         "of backward graph + all-reduce + optimiser graph; recorded in the line as dp_one_graph (same as "
                         "MKGNN_DP_ONE_GRAPH=1).  Not the default until it has run on more than one GPU: a collective that hangs "
                         "inside a replay cannot be caught")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check (no GPU needed): every rank prints its RANK / LOCAL_RANK / WORLD_SIZE as one JSON line and exits")
    ap.add_argument("--fresh-batches", type=int, default=16,
                    help="also time an epoch of this many distinct batches (one AID-1798 epoch at batch 4096 is 16) with the "
                         "receptive-field and index-plan build inside the timed region; 0 = skip")
    return ap.parse_args()


def assay_seed(assay):
    """Seed base of an assay's synthetic batches ('all9': the nine assays mixed, BASELINE configs[4])."""
    return int(assay) if str(assay).isdigit() else 9


def layer_algorithmic(plan, F, E, Ls, last):
    """Strict algorithmic bytes and flops of one KernelSetConv forward:
    x read once, the output written once, bond attributes, the index tensors (int64, as the ABI
    takes them), the kernel bank once, and the degree-4 coordinates in the last layer."""
    n = plan.n_atoms
    m = plan.n_slots
    K = sum(Ls)
    bank = sum(L * (F + d * F + d * E + 3 * d) for d, L in zip(range(1, 5), Ls))
    by = 4 * n * F + 4 * n * K + 4 * m * E + 8 * m + 8 * n + 4 * bank
    if last:
        by += 4 * plan.buckets[3].count * 15
    fl = 0
    for d, L in zip(range(1, 5), Ls):
        fl += plan.buckets[d - 1].count * L * (2 * F * (d * d + 1) + 2 * E * d * d)
    return by, fl


def backward_algorithmic(plan, F, E, Ls):
    """Algorithmic bytes / flops per launch of the backward's kernels for one KernelSetConv.
    Useful work only: of the d x d matrix entries of a pair the chosen order used d, so a pair costs 2 F (d + 1) flops
    towards the rows and the same (+ 2 E d for the bond supports) towards the bank -- the masked dense products the
    kernels issue are d times that for the support part."""
    n, m = plan.n_atoms, plan.n_slots
    pairs = sum(plan.buckets[d - 1].count * L for d, L in zip(range(1, 5), Ls))
    bank_x = sum(L * (d + 1) * F for d, L in zip(range(1, 5), Ls))
    bank_all = sum(L * ((d + 1) * F + d * E) + 4 for d, L in zip(range(1, 5), Ls))
    fl_rows = sum(plan.buckets[d - 1].count * L * 2 * F * (d + 1) for d, L in zip(range(1, 5), Ls))
    fl_bank = sum(plan.buckets[d - 1].count * L * (2 * F * (d + 1) + 2 * E * d) for d, L in zip(range(1, 5), Ls))
    through = sum(plan.buckets[d - 1].count * d * L for d, L in zip(range(1, 5), Ls))
    return {
        "coef_prepare_kernel": (16 * pairs + 4 * through + 8 * pairs, 8 * pairs),      # pair records + neighbours' dL/dh blocks in, {g, order} records out
        "kc_backward_rows_stream": (8 * pairs + 4 * (n + m) * F + 4 * bank_x, fl_rows),  # records in, contribution rows out, bank once
        "kc_backward_bank_stream": (4 * n * F + 8 * pairs + 32 * m + 8 * m + 8 * n + 4 * bank_all, fl_bank),   # x once, records, unit bond rows, indices, one slab's worth out
        "kc_backward_bank_reduce": (4 * bank_all, 4 * bank_all),
        "csr_rows_kernel<gather>": (4 * (n + m) * F + 4 * (n + m) + 8 * n * F, 2 * (n + m) * F),     # contribution rows + CSR in, x in, grad_x out
    }


def step_rows(Fn, h, plan, params, E, variant="auto"):
    """``h`` in the form the training step hands a layer its rows: pre-split (functional.presplit_rows) where the layer takes that
    (the default since round 6: DESIGN 4.1f), else as it is.  Returns (rows, "pre-split" | "fp32")."""
    try:
        if variant in ("auto", "mfma") and os.environ.get("MKGNN_ROWS_SPLIT", "1") != "0" \
                and Fn.rows_split_supported(plan, params, int(h.shape[1]), E, plan.n_atoms):
            return Fn.presplit_rows(h), "pre-split"
    except Exception:
        pass
    return h, "fp32"


def forward_kernel_source_sha16():
    """sha256 (first 16 hex digits) of the sources the streamed forward kernel is built from: a committed PMC figure names the one
    it was collected at (tools/collect_pmc.py), so a line can say when its `traffic` no longer belongs to the kernel it times."""
    import hashlib
    h = hashlib.sha256()
    for name in ("kgnn_fwd_stream.hip", "kgnn_split.h"):
        with open(os.path.join(REPO, "molkgnn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_traffic(names):
    """(HBM bytes per launch, where it comes from) of the first of these files under profiles/ that exists."""
    for name in names:
        path = os.path.join(REPO, "profiles", name)
        try:
            pmc = json.load(open(path))
            src = pmc.get("kernel_source_sha16")
            stale = "" if src is None else ("" if src == forward_kernel_source_sha16() else
                                            "; STALE: the forward kernel's sources have changed since (re-collect: tools/round_profiles.sh <tag> <commit> pmc)")
            return pmc["hbm_bytes_per_launch"], (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of commit "
                                                + pmc.get("commit", "?") + ")" + stale)
        except Exception:
            continue
    return None, None


def time_backward_kernels(lib, Fn, h, plan, params, E, variant, reps):
    """HIP-event durations (ms, mean over `reps` calls) of the five kernels of one N-hop layer's backward, each alone on the
    GPU (mkgnn_debug_time_backward keeps the call on one stream)."""
    import ctypes
    out = (ctypes.c_float * 5)()
    # (the rows as the step feeds them: pre-split where the layer takes that -- the leaf that asks for grad_x carries the form's tags)
    hs, form = step_rows(Fn, h, plan, params, E, variant)
    x = hs.detach().requires_grad_(True)
    if form == "pre-split":
        setattr(x, Fn._INV_ATTR, (getattr(hs, Fn._INV_ATTR)[0], x._version))
        Fn.mark_rows_split(x)
    wgt = torch.randn(h.shape[0], sum(int(p.shape[0]) for p in params[0::7]), device=h.device)
    acc, cnt = [0.0] * 5, [0] * 5
    lib.mkgnn_debug_time_backward(1)
    try:
        for r in range(reps + 2):
            for p_ in params:
                p_.grad = None
            x.grad = None
            hh = Fn.kernelsetconv(x, plan, False, params, E, variant, block_rows=True, propagate=True)
            (hh * wgt).sum().backward()
            if lib.mkgnn_debug_last_backward_ms(out) != 0:
                return None
            if r >= 2:
                for k in range(5):
                    if out[k] >= 0:
                        acc[k] += float(out[k]); cnt[k] += 1
    finally:
        lib.mkgnn_debug_time_backward(0)
        for p_ in params:
            p_.grad = None
    names = ("coef_prepare_kernel", "kc_backward_rows_stream", "kc_backward_bank_stream", "kc_backward_bank_reduce", "csr_rows_kernel<gather>")
    return {n_: (acc[k] / cnt[k] if cnt[k] else None) for k, n_ in enumerate(names)}


def exact_fp32_leg(args, model, opt, dev, batches, log):
    """The same step with the node-feature products on the exact fp32 matrix instructions (v_mfma_f32_16x16x4_f32, bit for bit an
    fmaf chain: rounds 1-4's arithmetic; MKGNN_FWD_SPLIT=0 MKGNN_BWD_SPLIT=0) instead of three fp16 ones on exactly split
    operands -- what the split buys, timed by the same clock in the same run (missing 4).  Graphs of the same
    resident batches, captured with the run-time switches set; the N-hop forward kernel alone by HIP events as in `roofline`."""
    import ctypes
    from molkgnn_amd import _lib
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.train import backward as train_backward
    lib = _lib.load()
    lib.mkgnn_debug_last_fused_forward_ms.restype = ctypes.c_float
    Fn.debug_set_products(0, 0)
    try:
        graphs = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for b in batches[:2]:
                model.zero_grad(set_to_none=True)
                loss_backward(model, b)
                if opt is not None:
                    opt.step()
            for b in batches:
                model.zero_grad(set_to_none=True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    loss_backward(model, b)
                    if opt is not None:
                        opt.step()
                graphs.append(g)
        torch.cuda.current_stream().wait_stream(side)
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        wins = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                graphs[i % len(graphs)].replay()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / args.steps)
        wins.sort()
        del graphs
        # the N-hop forward kernel alone, as in `roofline`
        b = batches[0]
        plan = plan_from_data(b)
        layer = model.gnn_model.gnn.layers[1]
        params, E = layer._bank_params("train", b.x)
        K_in = model.gnn_model.gnn.num_kernels(0)
        gen = torch.Generator(device=dev).manual_seed(1)
        h_store = torch.zeros(b.x.shape[0], K_in + (-K_in) % 4, device=dev)
        h_store[:, :K_in] = torch.rand(b.x.shape[0], K_in, generator=gen, device=dev) * 2 - 1
        h = h_store[:, :K_in]
        for _ in range(3):
            Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
        lib.mkgnn_debug_time_fused_forward(max(2, args.roofline_reps))
        samples = []
        for _ in range(9):
            Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
            samples.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
        lib.mkgnn_debug_time_fused_forward(0)
        fwd_ms = sorted(samples)[len(samples) // 2]
        return {"what": "the same step and the same N-hop forward kernel with the node-feature products on v_mfma_f32_16x16x4_f32 "
                        "(exact fp32: MKGNN_FWD_SPLIT=0 MKGNN_BWD_SPLIT=0), h between layers as fp32 rows",
                "ms_per_step": round(1e3 * wins[len(wins) // 2], 4), "ms_per_step_min": round(1e3 * wins[0], 4),
                "forward_kernel_ms": round(fwd_ms, 5),
                "products": {"forward": "v_mfma_f32_16x16x4_f32", "backward": "v_mfma_f32_16x16x4_f32", "rows": "fp32"}}
    finally:
        Fn.debug_set_products(-1, -1)


def small_batch_leg(args, model, opt, dev, log):
    """BASELINE configs[2] in the same JSON line (outside the headline's timed region): AID 435008 shape, batch 256 -- the
    step is launch-latency bound there; ms per step of the resident-batch replay and the N-hop forward kernel's roofline
    fraction at that size.  configs[0]'s batch of 16 (the reference's own, README.md:81) rides along.  Both through the
    per-operator kernels (the default) and through the molecule-resident one-launch step (molkgnn_amd.molecule,
    MKGNN_MOLECULE=1): `ms_per_step` is the default path's, `molecule_resident` the other's."""
    from molkgnn_amd import _lib
    from molkgnn_amd import functional as Fn
    from molkgnn_amd import molecule as Mol
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import backward as train_backward
    import ctypes
    lib = _lib.load()
    lib.mkgnn_debug_last_fused_forward_ms.restype = ctypes.c_float

    def replay_ms(batches):
        with torch.no_grad():
            for b in batches:
                model(b)
        graphs = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for b in batches[:2]:
                model.zero_grad(set_to_none=True)
                loss_backward(model, b)
                if opt is not None:
                    opt.step()
            for b in batches:
                model.zero_grad(set_to_none=True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    loss_backward(model, b)
                    if opt is not None:
                        opt.step()
                graphs.append(g)
        torch.cuda.current_stream().wait_stream(side)
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        steps, wins = 200, []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                graphs[i % 4].replay()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / steps)
        wins.sort()
        del graphs
        return wins

    def forward_ms(batches):
        """Forward only (eval mode, no gradient: scoring molecules), one hipGraph per resident batch."""
        model.eval()
        try:
            graphs = []
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for b in batches:
                    model.gnn_model(b)
                for b in batches:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        model.gnn_model(b)
                    graphs.append(g)
            torch.cuda.current_stream().wait_stream(side)
            for g in graphs:
                g.replay()
            torch.cuda.synchronize()
            wins = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(200):
                    graphs[i % 4].replay()
                torch.cuda.synchronize()
                wins.append((time.perf_counter() - t0) / 200)
            del graphs
            return sorted(wins)[2]
        finally:
            model.train()

    def eager_ms(batches, n=120):
        """The same step launched eagerly (no graph: the reference's own training-loop style), host overhead included."""
        for i in range(10):
            model.zero_grad(set_to_none=True)
            loss_backward(model, batches[i % 4])
            if opt is not None:
                opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            model.zero_grad(set_to_none=True)
            loss_backward(model, batches[i % 4])
            if opt is not None:
                opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    out = {}
    mode0 = Mol._MODE
    for assay, B in (("435008", 256), ("1798", 16)):
        batches = [make_batch(B, seed=assay_seed(assay) * 1000 + 700 + i, assay=assay).to(dev) for i in range(4)]
        try:
            Mol._MODE = "0"
            pwins = replay_ms(batches)
            pfwd = forward_ms(batches)
            peag = eager_ms(batches)
            Mol._MODE = "1"
            try:
                mwins = replay_ms(batches)
                mfwd = forward_ms(batches)
                meag = eager_ms(batches)
            except Exception as exc:
                mwins = mfwd = meag = None
                log(f"molecule-resident step unavailable ({type(exc).__name__}: {exc})")
            # the path a run takes by default at this batch size (molkgnn_amd.molecule: up to 32 molecules the one-launch step)
            # (a captured step: up to 32 molecules; an EAGER step above that takes it only for a batch marked data.resident = True
            # -- round 6: the choice is a function of the batch, not of how often it has been seen; eager_ms_per_step below
            # times both paths by forcing them)
            default_mol = mwins is not None and mode0 != "0" and B <= (Mol._MAX_MOLS_FORCED if mode0 == "1" else Mol._MAX_MOLS_AUTO)
            wins = mwins if default_mol else pwins
            mol = {"default_path": "molecule_resident" if default_mol else "per_operator",
                   "per_operator_ms_per_step": round(1e3 * pwins[2], 4),
                   "molecule_resident_ms_per_step": None if mwins is None else round(1e3 * mwins[2], 4),
                   "eager_ms_per_step": {"what": "the same step launched eagerly, host overhead included (no hipGraph); the default eager "
                                                 "path above 32 molecules is per_operator unless the batch carries resident = True",
                                         "per_operator": round(1e3 * peag, 4),
                                         "molecule_resident": None if meag is None else round(1e3 * meag, 4)},
                   "forward_only_ms": {"what": "MolKGNNNet.forward in eval mode, no gradient (scoring a batch), graph replay",
                                       "per_operator": round(1e3 * pfwd, 4),
                                       "molecule_resident": None if mfwd is None else round(1e3 * mfwd, 4)},
                   "molecule_resident": "prepare + ONE fwd/loss/bwd launch (a workgroup per chunk of whole molecules) + fixed-order "
                                        "reduction + AdamW: 4 launches per step; the default up to 32 molecules (MKGNN_MOLECULE)"}
        finally:
            Mol._MODE = mode0
        b = batches[0]
        plan = plan_from_data(b)
        layer = model.gnn_model.gnn.layers[1]
        params, E = layer._bank_params("train", b.x)
        K_in = model.gnn_model.gnn.num_kernels(0)
        h_store = torch.zeros(b.x.shape[0], K_in + (-K_in) % 4, device=dev)
        h_store[:, :K_in] = torch.rand(b.x.shape[0], K_in, device=dev) * 2 - 1
        h, _ = step_rows(Fn, h_store[:, :K_in], plan, params, E, args.variant)
        samples = []
        lib.mkgnn_debug_time_fused_forward(8)
        for r in range(7):
            Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
            if r >= 2:
                samples.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
        lib.mkgnn_debug_time_fused_forward(0)
        ms_f = sorted(samples)[len(samples) // 2]
        by, fl = layer_algorithmic(plan, K_in, E, layer.L, False)
        key = f"aid{assay}_b{B}"
        tr, tr_src = committed_traffic(("r06_forward_pmc_b256.json", "r05_forward_pmc_b256.json")) if (B == 256 and args.variant in ("auto", "mfma")) else (None, None)
        out[key] = {"forward_kernel_traffic": tr, "forward_kernel_traffic_source": tr_src, "forward_kernel_algorithmic_bytes": by,
                    "workload": f"AID {assay} shape, batch {B} ({b.x.shape[0]} atoms), fwd+bwd+AdamW, one hipGraph per resident batch",
                    "ms_per_step": round(1e3 * wins[2], 4), "ms_per_step_min": round(1e3 * wins[0], 4),
                    "value": round(B / wins[2], 1), "unit": "molecules/s",
                    "forward_kernel_ms": round(ms_f, 5), "forward_kernel_frac": round(by / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "paths": mol}
        log(f"small batch {key}: {1e3 * wins[2]:.4f} ms per step ({mol['default_path']}; per operator {mol['per_operator_ms_per_step']}, "
            f"molecule-resident {mol['molecule_resident_ms_per_step']}), forward kernel {1e3 * ms_f:.1f} us")
    return out


def host_cores():
    """CPU cores this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(seconds, assay):
    """Reference-faithful CPU restatement (oracle), forward + backward, batch 16 (README.md:81),
    on all host cores; bounded to about `seconds` of work."""
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import GNNModel
    from oracle import kgnn_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = GNNModel()
    state = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
             for k, v in model.state_dict().items()}
    gstate = {k[len("gnn_model."):]: v for k, v in state.items() if k.startswith("gnn_model.")}
    batches = [make_batch(16, seed=100 + i, assay=assay) for i in range(4)]
    done = 0
    t0 = time.perf_counter()
    while True:
        b = batches[done % len(batches)]
        emb = O.molkgnnnet(gstate, b, num_layers=3, training_bn=True, form="faithful")
        pred = emb @ state["ffn.weight"].T + state["ffn.bias"]
        loss = torch.nn.functional.binary_cross_entropy_with_logits(pred.view(-1), b.y)
        loss.backward()
        for v in state.values():
            v.grad = None
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds and done >= 2:
            break
    return {"value": round(16 * done / el, 2), "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"{done} steps of batch 16 ({16 * done} molecules), fwd+bwd, oracle/kgnn_oracle.py faithful form, "
                      f"torch {torch.__version__} CPU, {cores} threads"}


def fresh_batches_leg(args, model, opt, dev, log):
    """An epoch of DISTINCT batches through ONE captured graph (molkgnn_amd.padding): every batch is padded to the epoch's
    common shape with one inert molecule (as a loader would collate it), copied into static buffers, and the replayed
    graph builds the degree buckets (mkgnn_rf_count / mkgnn_rf_fill), the unit bond rows and the index plan
    (mkgnn_plan_build) and runs forward + backward + AdamW.  The copy into the static buffers, the receptive-field build
    and the plan build are INSIDE the timed region.  Reported next to the resident-replay value."""
    from molkgnn_amd import padding as P
    from molkgnn_amd.receptive_field import attach_receptive_fields
    from molkgnn_amd.synthetic import make_batch
    from molkgnn_amd.train import backward as train_backward
    nb, B = args.fresh_batches, args.batch_size
    raws = [make_batch(B, seed=assay_seed(args.assay) * 1000 + 500 + i, assay=args.assay, with_receptive_fields=False) for i in range(nb)]
    shape = P.fixed_shape([P.degree_histogram(r) for r in raws])
    padded = [P.pack(P.pad_batch(r, shape, B).to(dev)) for r in raws]      # (packed: one copy loads a batch)
    pad_atoms = sum(shape["atoms"] - int(p.n_valid_atoms) for p in padded) / nb
    # (the epoch's largest molecule, padding molecules included: what lets the captured step run the fused tail, readout.tail_loss)
    sb = P.StaticBatch(padded[0], max_mol_atoms=max(getattr(p_, "max_mol_atoms", 1 << 30) for p_ in padded),
                       max_mol_edges=max(getattr(p_, "max_mol_edges", 1 << 30) for p_ in padded))

    def step():
        attach_receptive_fields(sb.data, sizes=sb.data.bucket_sizes, overlap=True)
        model.zero_grad(set_to_none=True)
        loss = loss_backward(model, sb.data)
        if opt is not None:
            opt.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        model.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step()
    torch.cuda.current_stream().wait_stream(side)
    for k in range(min(3, nb)):                          # warm-up replays
        sb.load(padded[k]); g.replay()
    torch.cuda.synchronize()
    reps = max(1, int(math.ceil(0.1 / (nb * 1.2e-3))))   # epochs per window (>= 0.1 s); five windows, the median is reported
    wins = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for p in padded:
                sb.load(p)
                g.replay()
        torch.cuda.synchronize()
        wins.append(time.perf_counter() - t0)
    wins.sort()
    el = wins[len(wins) // 2]
    steps = reps * nb
    log(f"fresh batches: {nb} distinct batches x {reps} epochs x 5 windows, {1e3 * el / steps:.4f} ms per step "
        f"(min {1e3 * wins[0] / steps:.4f}, max {1e3 * wins[-1] / steps:.4f})")
    out = {"value": round(B * steps / el, 1), "unit": "molecules/s", "ms_per_step": round(1e3 * el / steps, 4),
           "ms_per_step_min": round(1e3 * wins[0] / steps, 4), "ms_per_step_max": round(1e3 * wins[-1] / steps, 4),
           "distinct_batches": nb, "epochs_per_window": reps, "windows": len(wins), "padding_atoms_per_batch": round(pad_atoms, 1),
           "fixed_shape": shape,
           "in_timed_region": "copy of the padded batch into the static buffers + ONE batch-agnostic hipGraph: receptive-field "
                              "build, unit bond rows, index plan (all HIP, no host round trip), fwd + bwd + AdamW"}
    # the same epoch from packed shards on the host (molkgnn_amd/shards.py): page cache -> pinned staging with the padding
    # made on the host, in the compact wire form (every bond once, byte-valued attributes, nothing derived) -> host-to-device
    # copy -> static buffers -> one more captured graph that starts by expanding the batch.  PCIe and the loader are inside.
    try:
        if os.environ.get("MKGNN_NO_SHARD_EPOCH"):
            raise RuntimeError("skipped (MKGNN_NO_SHARD_EPOCH)")
        import tempfile
        from molkgnn_amd import shards as S
        with tempfile.TemporaryDirectory() as d:
            paths = S.write_shards(d, raws)
            workers = int(os.environ.get("MKGNN_LOADER_WORKERS", 3))
            loader = S.ShardLoader(paths, B, device=dev, prefetch=3, workers=workers, fixed_shape=True, compact=True)
            if loader.shape != shape:
                raise RuntimeError("loader shape differs from the padded batches'")
            csb = P.CompactStaticBatch(shape, B, raws[0].x.shape[1], raws[0].p.shape[1], raws[0].edge_attr.shape[1], dev)

            def cstep():
                csb.expand()
                attach_receptive_fields(csb.data, sizes=csb.data.bucket_sizes, overlap=True)
                model.zero_grad(set_to_none=True)
                loss = loss_backward(model, csb.data)
                if opt is not None:
                    opt.step()
                return loss

            first = next(iter(loader))
            csb.load(first)
            side2 = torch.cuda.Stream()
            side2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side2):
                for _ in range(2):
                    cstep()
                model.zero_grad(set_to_none=True)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=side2):
                    cstep()
            torch.cuda.current_stream().wait_stream(side2)
            for cb in loader:                                # warm-up epoch (page cache, pinned buffers)
                csb.load(cb); g2.replay()
            torch.cuda.synchronize()
            ewins = []
            # MKGNN_SHARD_AHEAD=k (diagnostics): the host stays at most k steps in front of the GPU, as a training loop that reads
            # its loss would; 0 = as far as the staging ring lets it (measured over 80-step windows: 0.997 ms per step free or
            # with k = 4, 1.048 with k = 2)
            ahead = int(os.environ.get("MKGNN_SHARD_AHEAD", 0))
            marks = [None] * max(ahead, 1)
            # a timed window is ONE pass over the shard list repeated `passes` times (one loader start-up -- thread pool, the
            # first batch's staging and copy with nothing to overlap -- per window, as in an epoch of thousands of steps;
            # 16 batches per pass would make that start-up a tenth of every step)
            passes = int(os.environ.get("MKGNN_SHARD_PASSES", 5))
            timed = S.ShardLoader(paths * passes, B, device=dev, prefetch=3, workers=workers, fixed_shape=True, compact=True)
            if timed.shape != shape:
                raise RuntimeError("loader shape differs from the padded batches'")
            for _ in range(5):
                t0 = time.perf_counter()
                n = 0
                for cb in timed:
                    csb.load(cb)
                    if ahead > 0 and marks[n % ahead] is not None:
                        marks[n % ahead].synchronize()
                    g2.replay()
                    if ahead > 0:
                        marks[n % ahead] = torch.cuda.Event()
                        marks[n % ahead].record()
                    n += 1
                torch.cuda.synchronize()
                ewins.append((time.perf_counter() - t0) / max(n, 1))
            ewins.sort()
            out["shard_epoch"] = {"value": round(B / ewins[2], 1), "unit": "molecules/s", "ms_per_step": round(1e3 * ewins[2], 4),
                                  "ms_per_step_min": round(1e3 * ewins[0], 4), "ms_per_step_max": round(1e3 * ewins[-1], 4),
                                  "loader_workers": workers, "steps_per_window": nb * passes, "host_steps_ahead": ahead,
                                  "bytes_per_batch": int(csb.wire.numel()),
                                  "bytes_per_batch_expanded": int(sb.flat.numel()),
                                  "in_timed_region": "memory-mapped shard -> pinned staging (fixed-shape padding on the host, compact "
                                                     "wire form) -> host-to-device copy -> static buffers -> one graph: expand, "
                                                     "receptive fields, plan, fwd + bwd + AdamW; 5 windows, median"}
            log(f"shard epoch: {1e3 * ewins[2]:.4f} ms per step ({B / ewins[2] / 1e6:.2f} M molecules/s), {workers} loader workers, "
                f"{csb.wire.numel() / 1e6:.1f} MB per batch on the wire")
    except Exception as exc:                                 # (reported, not fatal: the headline does not depend on it)
        out["shard_epoch"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: start N fresh rank processes -- this
    process has made no GPU call and makes none -- as ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 bench.py <same arguments>``, pass rank 0's JSON line through and exit with the children's
    code.  No retry: a rank that fails fails the run."""
    import subprocess
    n = args.gpus
    if not args.dry_launch and not os.environ.get("MKGNN_ALLOW_SHARED_GPU"):
        have = torch.cuda.device_count()                 # (counts devices without creating a HIP context)
        if have < n:
            print(f"bench.py: --gpus {n} asked for {n} ranks, one per GPU, but this box has {have} GPU(s); "
                  f"refusing to run a {have}-GPU job labelled n_gpus={n} "
                  "(MKGNN_ALLOW_SHARED_GPU=1 MKGNN_DIST_BACKEND=gloo rehearses several ranks on one card)",
                  file=sys.stderr, flush=True)
            raise SystemExit(2)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # the host driver supports dmabuf IPC only (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    rc = subprocess.run(cmd, env=env).returncode
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])                 # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if args.dry_launch:
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world_size": world,
                          "master": f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}"}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path for the HIP kernels)")
    n_dev = torch.cuda.device_count()
    if world > n_dev and not os.environ.get("MKGNN_ALLOW_SHARED_GPU"):
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s) on this box (one rank per GPU; "
                         "MKGNN_ALLOW_SHARED_GPU=1 MKGNN_DIST_BACKEND=gloo rehearses several ranks on one card)")
    local_rank %= max(1, n_dev)                          # (rehearsals put several ranks on one card)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    from molkgnn_amd import _lib, dp
    from molkgnn_amd import functional as Fn
    from molkgnn_amd.plan import plan_from_data
    from molkgnn_amd.synthetic import ASSAY_SIZES, make_batch
    from molkgnn_amd.train import GNNModel, configure_optimizer, tune_torch_backends
    from molkgnn_amd.train import backward as train_backward
    _lib.load()
    if not os.environ.get("MKGNN_NO_TUNE"):
        tune_torch_backends()
    # MKGNN_BENCH_DP_PATH=1 (rehearsal on one GPU): take the N > 1 code path -- backward graph, RCCL all-reduce of the flat
    # gradient buffer, one shared optimiser graph -- with a process group of ONE rank
    dp_path = world > 1 or bool(os.environ.get("MKGNN_BENCH_DP_PATH"))
    want_one_graph = args.dp_one_graph or os.environ.get("MKGNN_DP_ONE_GRAPH") == "1"
    if want_one_graph:
        # a collective captured into the step's graph: the process group's watchdog thread must not query the events of a
        # work object recorded in a capturing stream (hipErrorCapturedEvent aborts the process on this stack: PyTorch 2.10,
        # RCCL 2.26); with these set before the group is created the one-rank rehearsal captures and replays
        for k in ("TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ENABLE_MONITORING", "TORCH_NCCL_CUDA_EVENT_CACHE", "TORCH_NCCL_BLOCKING_WAIT"):
            os.environ.setdefault(k, "0")
    dp.init_process_group_from_env(os.environ.get("MKGNN_DIST_BACKEND", "nccl"), force=dp_path, device=dev)   # nccl = RCCL over xGMI

    torch.manual_seed(1798)                       # same initial weights on every rank
    model = GNNModel(ffn_dropout_rate=float(os.environ.get("MKGNN_BENCH_FFN_DROPOUT", "0.25"))).to(dev)   # (diagnostics; 0.25 = the reference default)
    model.gnn_model.gnn.set_variant(args.variant)
    model.train()
    opt = None if args.no_optimizer else configure_optimizer(model, lr=1e-3, capturable=not args.no_graph)
    reducer = dp.FlatGradAllReduce(model.parameters(), dp.NEVER_TRAINED, [n for n, _ in model.named_parameters()])

    # resident batches: rank r owns batches r, r + world, ... of the global stream (weak scaling)
    n_mol_assay = ASSAY_SIZES.get(args.assay, 61832)
    nb = max(1, min(args.distinct_batches, math.ceil(n_mol_assay / args.batch_size)))
    batches = []
    for i in range(nb):
        b = make_batch(args.batch_size, seed=assay_seed(args.assay) * 1000 + i * world + rank, assay=args.assay).to(dev)
        batches.append(b)
    atoms = sum(b.x.shape[0] for b in batches) / nb
    with torch.no_grad():                                 # index plans (sorted CSRs) are part of the resident input
        for b in batches:
            model(b)
    torch.cuda.synchronize()

    def step(i):
        b = batches[i % nb]
        model.zero_grad(set_to_none=True)
        loss = loss_backward(model, b)
        reducer.reduce()
        if opt is not None:
            opt.step()
        return loss

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # One hipGraph per resident batch: forward + backward + optimiser are launch-bound at this model size
    # (51 kernels of 5-190 us), so the step is captured once and replayed.  The gradient all-reduce
    # (N > 1) stays outside the graph, between the backward graph and the optimiser.
    graphs = None
    flat_opt = False
    # (a rank whose batch lacks a degree: its bank's gradient slots stay zero in the flat buffer, the has-gradient flags
    # that travel with the all-reduce tell every rank's optimiser which banks had a gradient anywhere -- dp.py)
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(3):                   # warm every lazy initialisation on the capture stream
                    step(i)
                graphs = []
                from molkgnn_amd.optim import FusedAdamW as _FA
                fill_in_graph = isinstance(opt, _FA)
                if fill_in_graph and dp_path:            # (flag and zero tensors of every batch's gradient pattern are made
                    for bb in batches:                   # before the captures: a host-to-device copy cannot be captured)
                        model.zero_grad(set_to_none=True)
                        loss_backward(model, bb)
                        reducer.prepare_patterns([reducer.grads()])
                # N > 1: RCCL's watchdog thread polls its events with HIP calls of its own; in the default ("global")
                # capture mode such a call from another thread invalidates the capture
                cap = {"capture_error_mode": "thread_local"} if dp_path else {}
                # MKGNN_DP_ONE_GRAPH=1: the collective is captured too -- backward, copy into the flat buffer, RCCL all-reduce
                # and the optimiser (reading the flat views) are ONE graph per batch, as on one GPU.  Not the default: a
                # collective that hangs inside a replay cannot be caught; the two-graph step is.
                one_graph = dp_path and opt is not None and fill_in_graph and want_one_graph
                if one_graph:
                    for grp in opt.param_groups:
                        grp["grad_scale"] = 1.0 / world
                    opt.set_grad_active(reducer.active_flags())
                    # the communicator's first collective allocates and synchronises: not inside a capture
                    reducer.all_reduce_filled()
                    torch.cuda.synchronize()
                for i in range(nb):
                    model.zero_grad(set_to_none=True)
                    g_fb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_fb, stream=side, **cap):
                        # (loss -> backward as one unit: the fused tail's last reduction rides on the backward's helper stream)
                        static_loss = loss_backward(model, batches[i])
                        if opt is not None and not dp_path:
                            opt.step()
                        # this graph's gradient tensors: every captured graph writes into its own (p.grad names only the
                        # last capture's), and the all-reduce after it must work on exactly these.  With the flat-buffer
                        # optimiser the copy into the flat buffer is the graph's last node: outside, only the collective
                        own_grads = reducer.grads()
                        if opt is not None and dp_path and fill_in_graph:
                            reducer.fill(own_grads)
                        if one_graph:
                            reducer.all_reduce_filled()
                            for p_, v_ in zip(reducer.params, reducer.views):
                                p_.grad = v_
                            opt.step()
                    graphs.append([g_fb, None, static_loss, own_grads])
                if opt is not None and dp_path and not one_graph:
                    # N > 1: backward graph -> gradients summed over the ranks in the flat buffer -> ONE optimiser graph
                    # for all batches that reads the flat views and divides by the world size itself
                    from molkgnn_amd.optim import FusedAdamW
                    flat_opt = isinstance(opt, FusedAdamW)
                    for entry in graphs:
                        if flat_opt and entry is not graphs[0]:
                            entry[1] = graphs[0][1]
                            continue
                        if flat_opt:
                            for p_, v_ in zip(reducer.params, reducer.views):
                                p_.grad = v_
                            for grp in opt.param_groups:
                                grp["grad_scale"] = 1.0 / world
                            opt.set_grad_active(reducer.active_flags())
                            reducer.prepare_patterns([e_[3] for e_ in graphs])
                        else:                            # (PyTorch optimiser: averaged gradients copied back, one graph per batch)
                            for p_, g_ in zip(reducer.params, entry[3]):
                                p_.grad = g_
                        entry[1] = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(entry[1], stream=side, **cap):
                            opt.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        except Exception as exc:                     # capture is an optimisation, never a requirement
            log(f"hipGraph capture FAILED ({type(exc).__name__}: {str(exc).splitlines()[0]}); running eagerly -- the line says "
                "\"graph_replay\": false: a different execution mode from the documented headline")
            if os.environ.get("MKGNN_BENCH_DEBUG"):
                import traceback
                traceback.print_exc()
            try:                                     # (a failed capture leaves a sticky HIP error behind: read it away)
                torch.cuda.synchronize()
            except Exception:
                pass
            _lib.load().mkgnn_debug_clear_error()
            graphs = None
            flat_opt = False
            if opt is not None:
                for grp in opt.param_groups:
                    if "grad_scale" in grp:
                        grp["grad_scale"] = 1.0
            torch.cuda.synchronize()

    # (N > 1, flat-buffer optimiser: its ONE launch goes out as a plain kernel behind the collective -- a replayed graph of one node
    # starts 6 us later; measured on the one-GPU rehearsal of this path, 0.704 -> 0.698 ms.  MKGNN_DP_OPT_EAGER=0: the graph)
    opt_eager = flat_opt and os.environ.get("MKGNN_DP_OPT_EAGER", "1") != "0"
    if graphs is not None:
        def step(i):                                 # noqa: F811  (replay form of the same step)
            g_fb, g_opt, loss, static_grads = graphs[i % nb]
            g_fb.replay()
            if g_opt is not None:
                if flat_opt:
                    reducer.all_reduce_filled()          # (the copy into the flat buffer is the backward graph's last node)
                    if opt_eager:                        # ONE launch: as a plain kernel behind the collective, not a graph of one node
                        opt.step()
                        return loss
                else:
                    reducer.reduce(static_grads)
                g_opt.replay()
            return loss

    log(f"{nb} resident batches of {args.batch_size} molecules ({atoms:.0f} atoms each); "
        f"{'hipGraph replay' if graphs is not None else 'eager launches'}; warmup")
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    log("timing")

    def timed_window():
        """EXACTLY --steps steps between barrier + synchronize pairs; the maximum over ranks."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        own = el
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        own_windows.append(own)                          # (this rank's own clock for the window: per-rank figures below)
        return el

    own_windows = []

    # how many ranks the collective really spans (an all-reduce of ones), outside the timed region
    ranks_seen = None
    if dist.is_initialized():
        t = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(t.item())))

    first = timed_window()
    n_windows = args.windows if args.windows > 0 else max(5, min(200, int(math.ceil(0.5 / max(first, 1e-6)))))
    if world > 1:                                        # every rank must run the same number of windows
        t = torch.tensor([n_windows], dtype=torch.int64, device=dev)
        dist.broadcast(t, 0)
        n_windows = int(t.item())
    unsorted = [first] + [timed_window() for _ in range(n_windows - 1)]
    windows = sorted(unsorted)
    elapsed = windows[len(windows) // 2]                 # the median window is the reported one
    # N > 1, for whoever has to explain the scaling curve: every rank's own time for the reported window (the slowest sets
    # `value`), and the gradient all-reduce's own GPU time -- HIP events around the collective, one more untimed window
    per_rank = None
    allreduce_ms = None
    if dist.is_initialized():
        k_med = unsorted.index(elapsed)
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = own_windows[k_med]
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per = [1e3 * float(v) / args.steps for v in t.tolist()]
        per_rank = {"ms_per_step_min": round(min(per), 4), "ms_per_step_max": round(max(per), 4),
                    "ms_per_step_by_rank": [round(v, 4) for v in per],
                    "what": "each rank's own wall clock for the reported (median) window, barrier to barrier"}
        if graphs is not None and flat_opt:
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            dist.barrier()
            for i, (e0, e1) in enumerate(pairs):
                g_fb, g_opt, _, _ = graphs[i % nb]
                g_fb.replay()
                e0.record()
                reducer.all_reduce_filled()
                e1.record()
                g_opt.replay()
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
            t = torch.tensor([ts[len(ts) // 2], ts[0], ts[-1]], dtype=torch.float64, device=dev)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            allreduce_ms = {"median": round(float(tmax[0]), 4), "min": round(float(tmax[1]), 4), "max": round(float(tmax[2]), 4),
                            "bytes": reducer.nbytes,
                            "what": "HIP events around the flat-buffer all-reduce between the backward graph and the optimiser graph "
                                    "(includes waiting for the slowest rank's backward), max over ranks, one untimed window"}
    mols = args.batch_size * args.steps * world
    value = mols / elapsed
    # data-parallel sanity, outside the timed region: identical initial weights + averaged gradients + a deterministic
    # optimiser must leave every replica with the same parameters
    replicas_diff = None
    if world > 1 and opt is not None:
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in model.parameters()])
            ref = flat.clone()
            dist.broadcast(ref, 0)
            d = (flat - ref).abs().max().reshape(1)
            dist.all_reduce(d, op=dist.ReduceOp.MAX)
            replicas_diff = float(d.item())
        log(f"replicas: max |parameter - rank 0's| over all ranks after {args.warmup + args.steps} steps = {replicas_diff:g}")
    # ... and the BUFFERS (BatchNorm running statistics): every rank accumulated its own batches' statistics; averaged over
    # the ranks once here, as a training run would at a checkpoint (dp.BufferSync), then checked like the parameters
    buffers_before = buffers_after = None
    if dist.is_initialized():
        bsync = dp.BufferSync(model)
        buffers_before = bsync.max_abs_diff()
        bsync.average()
        buffers_after = bsync.max_abs_diff()
        log(f"buffers: max |buffer - rank 0's| {buffers_before:g} before the average over ranks, {buffers_after:g} after")

    # the collective part is over: check it, then every rank leaves the process group; rank 0 goes on alone with the
    # kernel measurements and the CPU baseline (the other ranks must not sit in a collective for those 30 s)
    dp_info = {}
    dp_failure = None
    if dist.is_initialized():
        dp_info = {"dp_ranks_seen": ranks_seen, "dp_backend": dist.get_backend(), "dp_one_graph": bool(want_one_graph),
                   "dp_per_rank": per_rank, "dp_allreduce_ms": allreduce_ms,
                   "dp_buffers": {"max_abs_diff_before_sync": buffers_before, "max_abs_diff": buffers_after,
                                  "sync": "BatchNorm running statistics averaged over the ranks once after the timed steps "
                                          "(dp.BufferSync.average; outside the timed region)"},
                   "dp_step": "one graph (collective captured)" if (graphs is not None and all(e_[1] is None for e_ in graphs))
                              else ("backward graph + all-reduce + optimiser graph" if graphs is not None else "eager")}
        if ranks_seen != world:
            dp_failure = f"the all-reduce spanned {ranks_seen} rank(s), not the {world} that --gpus asked for"
        elif replicas_diff is not None and replicas_diff > 0.0:
            dp_failure = f"replicas diverged: max |parameter - rank 0's| = {replicas_diff:g} after {args.warmup + args.steps} steps"
        elif buffers_after is not None and buffers_after > 0.0:
            dp_failure = f"buffers differ after the average over ranks: max |buffer - rank 0's| = {buffers_after:g}"
        dist.barrier()
        dist.destroy_process_group()
    if dp_failure is not None:
        if rank == 0:
            print(f"bench.py: data-parallel check FAILED: {dp_failure}; {value:.0f} molecules/s NOT reported", file=sys.stderr, flush=True)
        raise SystemExit(3)
    if rank != 0:
        return
    log(f"{value:.0f} molecules/s; measuring the forward kernels")
    out = None
    if rank == 0:
        # ---- roofline of the kernel-convolution forward, N-hop layer (F = 110), live HIP events on this stream
        b = batches[0]
        plan = plan_from_data(b)
        layer = model.gnn_model.gnn.layers[1]
        params, E = layer._bank_params("train", b.x)
        Ls = layer.L
        K_in = model.gnn_model.gnn.num_kernels(0)
        g = torch.Generator(device=dev).manual_seed(1)
        h_store = torch.zeros(b.x.shape[0], K_in + (-K_in) % 4, device=dev)
        h_store[:, :K_in] = torch.rand(b.x.shape[0], K_in, generator=g, device=dev) * 2 - 1
        h_fp32 = h_store[:, :K_in]
        # (the rows as the step feeds them: pre-split by their producer -- the kernel timed here is the one the step runs)
        h, rows_form = step_rows(Fn, h_fp32, plan, params, E, args.variant)
        import ctypes
        lib = _lib.load()
        lib.mkgnn_debug_last_fused_forward_ms.restype = ctypes.c_float
        saved_state = True    # the training configuration: permutation ids and scores are written too
        for _ in range(3):
            Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
        # (a) the whole forward call: row norms + bank preparation + output memset + the fused kernel
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(args.roofline_reps):
            Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
        ev1.record()
        torch.cuda.synchronize()
        ms_call = ev0.elapsed_time(ev1) / args.roofline_reps
        # (b) the dominant kernel alone (kc_forward_fused): HIP events recorded around its launch, on its stream
        # -- `roofline_reps` launches back to back between ONE event pair per sample (the kernel is idempotent); the median of
        # 9 samples.  An event pair around a single launch read 5-10 % above the rocprofv3 duration of the same kernel.
        samples, single = [], []
        if args.variant != "generic":                # (the generic kernels have no fused launch to bracket)
            lib.mkgnn_debug_time_fused_forward(max(2, args.roofline_reps))
            for _ in range(9):
                Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
                samples.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
            lib.mkgnn_debug_time_fused_forward(1)
            for _ in range(9):
                Fn.kernelsetconv_details(h, plan, False, params, E, args.variant, raw=True)
                single.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
            lib.mkgnn_debug_time_fused_forward(0)
        ms = sorted(samples)[len(samples) // 2] if samples and min(samples) > 0 else ms_call
        ms_single = sorted(single)[len(single) // 2] if single and min(single) > 0 else None
        by, fl = layer_algorithmic(plan, K_in, E, Ls, False)
        gbs = by / (ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed rocprofv3 PMC passes (same kernel, same workload; tools/pmc.sh -- counters
        # cannot be read from inside this process: the figure is the committed one, labelled with the commit it was taken at)
        traffic, traffic_source = (committed_traffic(("r06_forward_pmc.json", "r05_forward_pmc.json", "r04_forward_pmc.json"))
                                   if args.batch_size == 4096 and args.variant in ("auto", "mfma") else (None, None))
        # the other kernels of the N-hop layer (the bank gradient is the step's largest line), each alone on the GPU,
        # HIP events in this run; flops / bytes are the USEFUL ones (backward_algorithmic)
        kernels = [{"kernel": "kc_forward_stream<7, bf16 operands>" if args.variant == "bf16" else ("kc_forward_stream<7, pre-split rows>" if rows_form == "pre-split" else "kc_forward_stream<7>"), "ms_per_launch": round(ms, 5), "rows": rows_form, "algorithmic_bytes": by, "algorithmic_flops": fl,
                    "hbm_frac": round(gbs / HBM_PEAK_GBS, 5), "fp32_frac": round(fl / (ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5)}]
        try:
            # the 1-hop layer's forward (reference KernelLayer.py:21-37: F = 28, ~19 flop/B by the strict count -- at the fp32
            # ridge, the one KernelConv forward that HBM could bound): same kernel family, its KC = 2 instantiation
            layer0 = model.gnn_model.gnn.layers[0]
            params0, E0 = layer0._bank_params("train", b.x)
            F0 = int(b.x.shape[1])
            x0 = torch.zeros(b.x.shape[0], F0 + (-F0) % 4, device=dev)
            x0[:, :F0] = torch.randn(b.x.shape[0], F0, generator=g, device=dev)
            x0, _ = step_rows(Fn, x0[:, :F0], plan, params0, E0, args.variant)
            s0 = []
            if args.variant != "generic":
                for _ in range(3):
                    Fn.kernelsetconv_details(x0, plan, False, params0, E0, args.variant, raw=True)
                lib.mkgnn_debug_time_fused_forward(max(2, args.roofline_reps))
                for _ in range(9):
                    Fn.kernelsetconv_details(x0, plan, False, params0, E0, args.variant, raw=True)
                    s0.append(float(lib.mkgnn_debug_last_fused_forward_ms()))
                lib.mkgnn_debug_time_fused_forward(0)
            if s0 and min(s0) > 0:
                ms0 = sorted(s0)[len(s0) // 2]
                by0, fl0 = layer_algorithmic(plan, F0, E0, layer0.L, False)
                t0_, src0 = (committed_traffic(("r06_forward_pmc_1hop.json", "r05_forward_pmc_1hop.json")) if args.batch_size == 4096 and args.variant in ("auto", "mfma")
                             else (None, None))
                kernels.append({"kernel": "kc_forward_stream<2, bf16 operands>" if args.variant == "bf16" else "kc_forward_stream<2>",
                                "what": "KernelSetConv forward of the 1-hop layer (F=28, K=110), training configuration",
                                "ms_per_launch": round(ms0, 5), "algorithmic_bytes": by0, "algorithmic_flops": fl0,
                                "hbm_frac": round(by0 / (ms0 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                "fp32_frac": round(fl0 / (ms0 * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5),
                                "traffic": t0_, "traffic_source": src0})
        except Exception as exc:
            log(f"1-hop forward timing unavailable ({type(exc).__name__}: {exc})")
        try:
            if args.variant in ("auto", "mfma"):
                bt = time_backward_kernels(lib, Fn, h_fp32, plan, params, E, args.variant, max(5, args.roofline_reps // 2))
                alg = backward_algorithmic(plan, K_in, E, Ls)
                for name, t_ms in (bt or {}).items():
                    if t_ms:
                        b_, f_ = alg[name]
                        kernels.append({"kernel": name, "ms_per_launch": round(t_ms, 5), "algorithmic_bytes": b_, "algorithmic_flops": f_,
                                        "hbm_frac": round(b_ / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                        "fp32_frac": round(f_ / (t_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5)})
        except Exception as exc:
            log(f"backward kernel timing unavailable ({type(exc).__name__}: {exc})")
        roofline = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": ("kc_forward_stream<7, bf16 operands>" if args.variant == "bf16" else
                               ("kc_forward_stream<7, 3>" if rows_form == "pre-split" else "kc_forward_stream<7, 2>"))
                              + ": one launch = KernelSetConv forward of one N-hop layer (F=110, K=110), "
                              "all four degree buckets, training configuration (saves the pair records)"
                              + ("; atom rows pre-split by their producer, as in the step" if rows_form == "pre-split" else ""),
                    "ms_per_launch": round(ms, 5), "ms_whole_forward_call": round(ms_call, 5),
                    "timing": f"HIP events on the kernel's stream around {max(2, args.roofline_reps)} back-to-back launches, median of 9 samples",
                    "ms_per_launch_single_bracket": None if ms_single is None else round(ms_single, 5),
                    "algorithmic_bytes": by, "algorithmic_flops": fl,
                    "fp32_tflops": round(fl / (ms * 1e-3) / 1e12, 3),
                    "fp32_vector_frac": round(fl / (ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 5),
                    # how the node-feature products are computed: fp32_frac / fp32_vector_frac are the rate of
                    # useful fp32 multiply-adds against the fp32 peak, not a pipe utilisation -- the split products run at
                    # the fp16 matrix rate, three instructions of 8 cycles in place of four of 32
                    "products": products_mode(args.variant),
                    "kernels": kernels}
        out = {"metric": "molecules/sec fwd+bwd, 3-layer MolKGNN on AID 1798", "value": round(value, 1),
               "unit": "molecules/s", "n_gpus": (ranks_seen if ranks_seen is not None else world), "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4),
               "windows": {"n": len(windows), "steps_each": args.steps, "statistic": "median",
                           "ms_per_step_min": round(1e3 * windows[0] / args.steps, 4),
                           "ms_per_step_max": round(1e3 * windows[-1] / args.steps, 4),
                           "timed_seconds_total": round(sum(windows), 4)},
               "graph_replay": graphs is not None,       # False: the capture failed (or --no-graph) and the steps were launched eagerly
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16 dot products, f32 otherwise" if args.variant == "bf16" else "f32",
               # (f32 in, f32 out, f32 accumulation; the node-feature products are formed from exact fp16 halves of the f32
               # operands unless MKGNN_FWD_SPLIT=0 / MKGNN_BWD_SPLIT=0 -- roofline.products; fp32-grade against float64)
               "dtype_note": products_mode(args.variant)["forward"],
               "data": "synthetic",
               **({"dp_replicas_max_abs_diff": replicas_diff} if replicas_diff is not None else {}),
               **dp_info,
               "config": {"workload": f"{'all nine assays mixed' if args.assay == 'all9' else 'AID ' + args.assay} full set shape ({n_mol_assay} molecules, ~25 atoms / ~53 directed "
                                      f"edges each), 3 layers, hidden_dim 32, kernels 10/20/30/50 per degree, "
                                      f"batch {args.batch_size} molecules per GPU ({atoms:.0f} atoms), "
                                      f"{nb} resident batches cycled, fwd+bwd"
                                      + ("" if args.no_optimizer else "+AdamW") + (", grad all-reduce" if world > 1 else "")
                                      + (", one hipGraph per batch" if graphs is not None else ", eager launches"),
                          "variant": args.variant, "batch_size_per_gpu": args.batch_size,
                          "parallelism": f"dp{world}"},
               "roofline": roofline}
        log(f"forward kernel {ms:.4f} ms ({ms_call:.4f} ms whole call), {gbs:.1f} GB/s algorithmic")
        # the legs below train a model of their own at N > 1 (this one's optimiser reads the flat all-reduce buffer)
        leg_model, leg_opt = model, opt
        if dp_path and (args.fresh_batches > 0 or not os.environ.get("MKGNN_NO_SMALL_BATCH")):
            torch.manual_seed(1798)
            leg_model = GNNModel(ffn_dropout_rate=float(os.environ.get("MKGNN_BENCH_FFN_DROPOUT", "0.25"))).to(dev)
            leg_model.gnn_model.gnn.set_variant(args.variant)
            leg_model.train()
            leg_opt = None if args.no_optimizer else configure_optimizer(leg_model, lr=1e-3, capturable=not args.no_graph)
        if args.variant in ("auto", "mfma") and not os.environ.get("MKGNN_NO_EXACT_LEG"):
            try:
                out["exact_fp32_instructions"] = exact_fp32_leg(args, leg_model, leg_opt, dev, batches, log)
            except Exception as exc:                         # (reported, not fatal)
                out["exact_fp32_instructions"] = {"error": f"{type(exc).__name__}: {exc}"}
        if args.fresh_batches > 0:
            try:
                out["fresh_batches"] = fresh_batches_leg(args, leg_model, leg_opt, dev, log)
            except Exception as exc:                         # (reported, not fatal)
                out["fresh_batches"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not os.environ.get("MKGNN_NO_SMALL_BATCH"):
            try:
                out["small_batch"] = small_batch_leg(args, leg_model, leg_opt, dev, log)
            except Exception as exc:                         # (reported, not fatal)
                out["small_batch"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.assay)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
