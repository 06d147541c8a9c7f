"""Throughput of the packed-shard loader (molkgnn_amd/shards.py) on the GPU box: molecules/s of (a) staging + host-to-device
copy + device-side index rebuild alone, (b) the same followed by the HIP receptive-field builder and index plan, i.e. a
batch ready for the model.  tools/shard_loader_probe.py [--molecules 65536] [--batch-size 4096]"""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molkgnn_amd import shards as S                                    # noqa: E402
from molkgnn_amd.plan import plan_from_data                             # noqa: E402
from molkgnn_amd.receptive_field import attach_receptive_fields         # noqa: E402
from molkgnn_amd.synthetic import make_batch                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--molecules", type=int, default=65536)
ap.add_argument("--batch-size", type=int, default=4096)
ap.add_argument("--assay", default="all9")
ap.add_argument("--workers", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    paths = S.write_shards(d, [make_batch(args.molecules // 4, seed=50 + i, assay=args.assay, with_receptive_fields=False) for i in range(4)])
    size = sum(os.path.getsize(p) for p in paths)
    print(f"wrote {len(paths)} shards, {size / 1e6:.1f} MB for {args.molecules} molecules ({size / args.molecules:.0f} B/molecule) "
          f"in {time.perf_counter() - t0:.1f} s (generation included)")
    for what in ("copy", "copy+rf+plan"):
        for rep in range(3):
            loader = S.ShardLoader(paths, args.batch_size, device=dev, prefetch=3, workers=args.workers)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for b in loader:
                if what != "copy":
                    attach_receptive_fields(b)
                    plan_from_data(b).build_hip()
                n += b.num_graphs
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        print(f"{what:14s}: {n / el / 1e6:.2f} M molecules/s ({1e3 * el / len(loader):.3f} ms per batch of {args.batch_size})")
        del loader, b                      # (pinned staging buffers must go before the interpreter tears the runtime down)
