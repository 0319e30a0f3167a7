"""CPU, gloo, world_size 2: the data-parallel exchange step (flat-buffer bucketed gradient all-reduce), parameter
broadcast and tile sharding.  Runs without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from amyloid_yolo_paper_amd.parallel import FlatGradReducer, broadcast_parameters, shard_indices


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
