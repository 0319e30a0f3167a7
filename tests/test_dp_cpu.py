"""CPU, gloo, world_size 2: the data-parallel exchange step (flat-buffer bucketed gradient all-reduce), parameter
broadcast and tile sharding.  Runs without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from amyloid_yolo_paper_amd.parallel import FlatGradReducer, broadcast_parameters, shard_indices, shard_indices_equal


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.LeakyReLU(0.1),
                               torch.nn.Conv2d(8, 5, 1), torch.nn.Flatten(), torch.nn.Linear(5 * 16, 7))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _tiny_model(100 + rank)              # different init per rank ...
        broadcast_parameters(model, src=0)           # ... made identical
        ref = _tiny_model(100)
        same = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), ref.state_dict().values()))
        red = FlatGradReducer(model.parameters(), n_buckets=3)
        n_params = sum(p.numel() for p in model.parameters())
        assert red.flat.numel() == n_params and red.views_intact()
        assert red.buckets[0][0] == 0 and red.buckets[-1][1] == n_params
        assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))
        # two accumulated micro-batches per rank, different data per rank (tiles sharded by image)
        torch.manual_seed(7)
        data = torch.randn(8, 3, 4, 4)
        mine = shard_indices(8, rank, world)
        for k in range(2):
            x = data[mine[2 * k:2 * k + 2]]
            model(x).square().sum().backward()       # autograd accumulates into the flat views in place
        assert red.views_intact()
        local = red.flat.clone()
        nbytes = red.all_reduce()
        assert nbytes == n_params * 4
        # reference: average over ranks of each rank's locally accumulated gradient
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        ok = torch.allclose(red.flat, want, rtol=1e-6, atol=1e-7)
        # .grad of every parameter sees the reduced values (views), optimizer step keeps replicas identical
        opt = torch.optim.Adam(model.parameters())
        opt.step()
        opt.zero_grad(set_to_none=False)
        assert red.views_intact() and float(red.flat.abs().sum()) == 0.0
        flat_params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        ps = [torch.zeros_like(flat_params) for _ in range(world)]
        dist.all_gather(ps, flat_params)
        identical = all(torch.equal(ps[0], t) for t in ps)
        q.put((rank, same, ok, identical))
    finally:
        dist.destroy_process_group()


class _Layered(torch.nn.Module):
    """parameters named like Darknet's: module_list.{i}.conv_{i}.weight, module_list.{i}.batch_norm_{i}.weight|bias"""

    def __init__(self, seed, n_layers=6):
        super().__init__()
        torch.manual_seed(seed)
        self.module_list = torch.nn.ModuleList()
        for i in range(n_layers):
            seq = torch.nn.Sequential()
            seq.add_module(f"conv_{i}", torch.nn.Conv2d(4, 4, 3, padding=1, bias=False))
            seq.add_module(f"batch_norm_{i}", torch.nn.BatchNorm2d(4))
            self.module_list.append(seq)


def _overlap_worker(rank, world, port, q):
    """the exchange overlapped with the backward walk: the engine reports finished layers deepest first
    (model._grad_ready(layer)), a bucket goes out when its shallowest layer is done; same result as the plain exchange,
    same number of collectives on every rank for an odd-sized dataset"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _Layered(5)
        red = FlatGradReducer(model.parameters(), n_buckets=3).attach(model)
        assert model._grad_ready == red.layer_done and len(red.bucket_first_layer) == len(red.buckets)
        assert red.bucket_first_layer == sorted(red.bucket_first_layer) and red.bucket_first_layer[0] == 0
        names = [n for n, _ in model.named_parameters()]
        g = torch.Generator().manual_seed(100 + rank)
        order_ok = True
        # an odd-sized dataset: 17 tiles, 2 ranks, batch 4 -> equal shards of 9 -> 3 batches on EVERY rank
        mine = shard_indices_equal(17, rank, world)
        n_batches = (len(mine) + 3) // 4
        exchanges = 0
        for batches_done in range(n_batches):
            step_now = batches_done % 2 == 0
            if step_now:
                red.begin()
            issued_at = {}
            for layer in range(len(model.module_list) - 1, -1, -1):      # the backward walk, deepest layer first
                for n, p in zip(names, red.params):
                    if n.startswith(f"module_list.{layer}."):
                        p.grad.add_(torch.randn(p.shape, generator=g))      # kernels ADD into the flat views
                before = set(red._handles)
                model._grad_ready(layer)
                for k in set(red._handles) - before:
                    issued_at[k] = layer
            local = red.flat.clone()                                        # (gloo reduces in place: clone before the handles finish?)
            model._grad_ready(-1)
            if step_now:
                # a bucket is issued exactly when the walk reaches its shallowest layer -- never earlier
                order_ok &= all(issued_at.get(k, -1) == red.bucket_first_layer[k] for k in range(len(red.buckets)))
                red.all_reduce(average=True)
                exchanges += 1
                red.zero()
            else:
                order_ok &= not red._handles                                # not armed: nothing goes out mid-accumulation
        # equality with the plain exchange on fresh, known gradients
        red.begin()
        for k, p in enumerate(red.params):
            p.grad.fill_(float(rank + 1) * (k + 1))
        for layer in range(len(model.module_list) - 1, -1, -1):
            model._grad_ready(layer)
        red.all_reduce(average=True)
        want = torch.cat([torch.full((p.numel(),), (1 + 2) / 2.0 * (k + 1)) for k, p in enumerate(red.params)])
        q.put((rank, order_ok, exchanges, n_batches, bool(torch.allclose(red.flat, want))))
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({(ex, nb) for _, _, ex, nb, _ in res}) == 1, f"ranks disagree on batches / exchanges: {res}"
    for rank, order_ok, exchanges, n_batches, equal in res:
        assert order_ok, f"rank {rank}: a bucket went out before its shallowest layer was done (or while not armed)"
        assert exchanges == 2 and n_batches == 3
        assert equal, f"rank {rank}: overlapped exchange != mean of the ranks' gradients"


def test_equal_shards_for_training():
    for n in (1, 7, 16, 17, 200):
        for world in (1, 2, 3, 8):
            shards = [shard_indices_equal(n, r, world) for r in range(world)]
            assert len({len(s) for s in shards}) == 1                      # same number of tiles -> same number of batches
            assert set(i for s in shards for i in s) == set(range(n))       # every tile is seen
            assert sum(len(s) for s in shards) - n < world                  # at most world-1 tiles wrap around


def test_flat_grad_allreduce_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, ok, identical in res:
        assert same, f"rank {rank}: broadcast did not equalise parameters"
        assert ok, f"rank {rank}: all-reduce result != mean of local gradients"
        assert identical, f"rank {rank}: replicas diverged after the optimiser step"


def test_shard_indices_cover_every_tile_once():
    for n in (0, 1, 7, 16, 200):
        for world in (1, 2, 3, 8):
            shards = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for s in shards for i in s)
            assert flat == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_reducer_single_process_is_noop():
    m = _tiny_model(1)
    red = FlatGradReducer(m.parameters(), n_buckets=4)
    m(torch.randn(2, 3, 4, 4)).sum().backward()
    before = red.flat.clone()
    assert red.all_reduce() == 0 and torch.equal(before, red.flat)


# ---- bench.py's own launcher: `python bench.py --gpus N` from a plain start ---------------------------------------

def _run_bench(*args, timeout=240):
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, os.path.join(repo, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_launch_starts_one_rank_per_gpu_and_forwards_rank0():
    """world 2 over gloo through the launcher: both ranks rendezvous on 127.0.0.1, exactly one JSON line comes back on stdout"""
    import json
    r = _run_bench("--gpus", "2", "--launch_probe")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert (j["world"], j["n_gpus"], j["ranks_seen"], j["sum_ranks"], j["sum_local_ranks"], j["master"]) == (2, 2, 2, 1.0, 1.0, "127.0.0.1")


def test_bench_self_launch_fails_cleanly_in_the_children_without_a_gpu():
    """on a CPU-only box the launcher itself never touches torch.cuda: the CHILDREN report the missing device, the launcher stops the
    other rank and returns non-zero; no JSON line, no retry"""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a HIP device")
    r = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0
    assert "needs a HIP device" in r.stderr and "rank 0 of 2" in r.stderr + r.stdout
    assert "stopping the other ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_under_torch_distributed_run_keeps_the_launchers_world():
    """the driver's form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (WORLD_SIZE set: no self-launch)"""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--launch_probe"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["world"] == 2 and j["ranks_seen"] == 2
