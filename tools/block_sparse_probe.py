"""the block-sparse N-hop forward, one degree (d = 2), forward only -- far enough to keep or kill it.

An N-hop layer's input rows are h[j] = sum over j's in-neighbours t of sim[t] (KernelLayer.py:119-123), each sim[t] non-zero
only in the column block of t's degree (10 / 20 / 30 / 50 of 110 columns), and the cosine's numerator is linear in h:
    h[j] . s_hat[r] = sum_t sim[t, block(t)] . s_hat[r, block(t)].
Taking the products on the BLOCK rows and summing over the in-neighbours afterwards is exactly what the block-row readout
kernels already do for lin1 (readout.readout_blocks: block_project_mfma_kernel, then the propagate step on the projected
rows).  So the prototype of the degree-2 bank needs no new kernel: lin1 := the 60 unit rows of the degree-2 bank (40 supports
+ 20 centres; 64 with padding), and `pre` of mkgnn_readout_blocks_forward is S[j][r] = h[j] . s_hat[r] for every atom.

Reported: the two kernels' time (HIP events around the call minus the pooling kernel it cannot skip, timed separately), S against
h @ s_hat^T in float64 on a sample, and -- the comparison asked for -- the streamed forward kernel on a batch that has ONLY
its degree-2 bucket (the same atoms, the same kernels: kc_forward_stream<7>'s degree-2 share).
    python3 tools/block_sparse_probe.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import _lib                         # noqa: E402
from molkgnn_amd import functional as Fn             # noqa: E402
from molkgnn_amd import readout as R                 # noqa: E402
from molkgnn_amd.kernels import KernelSetConv       # noqa: E402
from molkgnn_amd.plan import plan_from_data, plan_from_lists   # noqa: E402
from molkgnn_amd.synthetic import make_batch        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
b = make_batch(B, seed=1798000).to(dev)
b.num_graphs = B
plan = plan_from_data(b)
n = b.x.shape[0]
layer0 = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=28, edge_attr_dim=7).to(dev)
layer1 = KernelSetConv(10, 20, 30, 50, D=3, node_attr_dim=110, edge_attr_dim=7).to(dev)
p0, E = layer0._bank_params("train", b.x)
p1, _ = layer1._bank_params("train", b.x)
with torch.no_grad():
    sim = Fn.kernelsetconv(b.x, plan, False, p0, E, "auto", block_rows=True)        # block rows of the 1-hop layer
    h = Fn.propagate_add(sim, plan, out_pad=2)                                        # the dense N-hop input [n, 110]
    # the degree-2 bank's unit rows as a "lin1": rows (b, l) = supports, then centres; 64 x 110
    xs, xc = p1[7 + 1].detach(), p1[7 + 0].detach()                                   # x_support [20, 2, 110], x_center [20, 110]
    rows = torch.cat([xs[:, 0], xs[:, 1], xc], 0)
    U2 = torch.zeros(64, 110, device=dev)
    U2[:60] = rows / rows.norm(dim=1, keepdim=True).clamp_min(1e-8)
    w2 = torch.zeros(32, 64, device=dev)
    seg = R.molecule_segments(b.batch, B)
    hs = int(lib.mkgnn_readout_hidden_stride(64))
    z = torch.empty((n, hs), device=dev); pre = torch.empty((n, hs), device=dev)
    pooled = torch.empty((B, hs), device=dev); out = torch.empty((B, 32), device=dev)
    prm = R._params(U2, None, w2, None)
    rowptr, col = plan.csr_in

    def block_sparse():
        _lib.check(lib.mkgnn_readout_blocks_forward(
            prm, sim.data_ptr(), int(sim.stride(0)), _lib.Int32x4(10, 20, 30, 50), R._sel_buckets(plan), n, rowptr.data_ptr(),
            col.data_ptr(), seg.mol_ptr.data_ptr(), seg.size, None, z.data_ptr(), pre.data_ptr(), pooled.data_ptr(), None,
            out.data_ptr(), 32, _lib.stream_ptr(dev)), "mkgnn_readout_blocks_forward")

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    t_all = timed(block_sparse)
    # correctness on a sample: S[j][r] = h[j] . s_hat[r]
    block_sparse()
    torch.cuda.synchronize()
    idx = torch.randint(0, n, (2000,), device=dev)
    ref = (h[idx, :110].double() @ U2[:60].double().T)
    err = float((pre[idx, :60].double() - ref).abs().max())
    # the streamed forward on the degree-2 bucket alone (same atoms, same 20 kernels)
    empty_l = torch.zeros(0, dtype=torch.long, device=dev)
    empty_f = torch.zeros(0, device=dev)
    bk = plan.buckets[1]
    lists = lambda v2, e: [e, v2, e, e]                                                # noqa: E731
    plan2 = plan_from_lists(n, lists(bk.p_focal, empty_f), lists(bk.nei_p, empty_f), lists(bk.e_nei, empty_f),
                            lists(bk.sel, empty_l), lists(bk.nei, empty_l), b.edge_index)
    hv = h[:, :110]
    t_d2 = timed(lambda: Fn.kernelsetconv_details(hv, plan2, False, p1, E, "auto", raw=True))
    t_full = timed(lambda: Fn.kernelsetconv_details(hv, plan, False, p1, E, "auto", raw=True))
    lib.mkgnn_debug_last_fused_forward_ms.restype = __import__("ctypes").c_float
    lib.mkgnn_debug_time_fused_forward(8)
    Fn.kernelsetconv_details(hv, plan2, False, p1, E, "auto", raw=True)
    k_d2 = float(lib.mkgnn_debug_last_fused_forward_ms()) * 1e3
    Fn.kernelsetconv_details(hv, plan, False, p1, E, "auto", raw=True)
    k_full = float(lib.mkgnn_debug_last_fused_forward_ms()) * 1e3
    lib.mkgnn_debug_time_fused_forward(0)
print(f"batch {B}: {n} atoms, {bk.count} of degree 2")
print(f"block-sparse S for the degree-2 bank (project on block rows + propagate of 64-wide rows + the pooling kernel the entry "
      f"point cannot skip): {t_all:.1f} us per call; max |S - h s_hat^T| on 2000 atoms x 60 rows = {err:.2e}")
print(f"streamed forward kernel, degree-2 bucket only: {k_d2:.1f} us (whole call {t_d2:.1f}); all four degrees: {k_full:.1f} us (whole call {t_full:.1f})")
print("(the block-sparse form still needs: the dense h for 1 / |h| -- the propagate pass stays --, and the pair epilogue's gather "
      f"of S rows: {bk.count} atoms x 3 rows x 240 B = {bk.count * 3 * 240 / 1e6:.1f} MB)")
