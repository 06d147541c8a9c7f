"""``bench.py --gpus N`` starts its own ranks: the launcher, checked without a GPU, and the
N > 1 step structure (backward -> fill -> all-reduce of the flat buffer -> optimiser on the flat views) on two gloo ranks."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MKGNN_ALLOW_SHARED_GPU")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_n_launches_n_ranks_with_their_own_environment():
    """--gpus 2 without a torchrun environment: two fresh rank processes, each with its RANK / LOCAL_RANK / WORLD_SIZE."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert sorted(l["local_rank"] for l in lines) == [0, 1]
    assert all(l["world_size"] == 2 and l["dry_launch"] for l in lines)
    assert all(l["master"].startswith("127.0.0.1:") for l in lines)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr


def test_gpus_n_on_a_box_with_fewer_gpus_fails_loudly():
    """Never a 1-GPU number labelled n_gpus = N: asking for more ranks than the box has GPUs is an error with a message,
    before anything is launched (this container has no GPU at all)."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has two GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "--gpus 2" in r.stderr and "GPU(s)" in r.stderr and "refusing" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())     # no JSON line


def test_launcher_rank_count_must_agree_with_gpus():
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step_worker(rank, world, port, out_dir):
    """bench.py's N > 1 step on the CPU: `backward graph` (here: autograd) ends with fill(own gradients); outside it only
    the collective; the optimiser reads the flat views, divides by the world size and skips parameters no rank had a
    gradient for."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from molkgnn_amd import dp
    assert dp.init_process_group_from_env("gloo") == world
    torch.manual_seed(11)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    unused = torch.nn.Parameter(torch.ones(3))              # a parameter nobody has a gradient for
    params = list(net.parameters()) + [unused]
    red = dp.FlatGradAllReduce(params)
    flags = red.active_flags()
    g = torch.Generator().manual_seed(50 + rank)
    lr = 0.1
    seen = torch.ones(1)
    dist.all_reduce(seen)                                   # bench.py's dp_ranks_seen
    for step in range(3):
        x, y = torch.randn(8, 6, generator=g), torch.randn(8, 1, generator=g)
        for p in params:
            p.grad = None
        ((net(x) - y) ** 2).mean().backward()
        red.fill(red.grads())                               # last node of the backward graph
        red.all_reduce_filled()                             # the only thing outside a graph
        with torch.no_grad():                               # the optimiser graph: flat views, 1 / world, has-gradient flags
            for p, v in zip(red.params, red.views):
                if float(flags[p]) > 0:
                    p -= lr * v / world
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    torch.save({"flat": flat, "seen": int(seen.item()), "unused_flag": float(flags[unused])}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_n_rank_step_structure_keeps_replicas_identical(tmp_path):
    port = _free_port()
    mp.spawn(_step_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["seen"] == r1["seen"] == 2
    assert torch.equal(r0["flat"], r1["flat"])              # bench.py's dp_replicas_max_abs_diff == 0
    assert r0["unused_flag"] == 0.0 and torch.equal(r0["flat"][-3:], torch.ones(3))
    # the same three steps in one process on the union of the two ranks' batches
    torch.manual_seed(11)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    gens = [torch.Generator().manual_seed(50 + r) for r in range(2)]
    for step in range(3):
        grads = None
        for g in gens:
            x, y = torch.randn(8, 6, generator=g), torch.randn(8, 1, generator=g)
            net.zero_grad()
            ((net(x) - y) ** 2).mean().backward()
            cur = [p.grad.clone() for p in net.parameters()]
            grads = cur if grads is None else [a + b for a, b in zip(grads, cur)]
        with torch.no_grad():
            for p, gsum in zip(net.parameters(), grads):
                p -= 0.1 * gsum / 2
    want = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(r0["flat"][:-3], want, atol=1e-6)
