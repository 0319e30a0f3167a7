"""Data parallelism over tiles: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  The reference has no multi-GPU code at all: this is new.

* Inference shards by image: `shard_indices` deals tile indices round-robin to ranks; no data-path collective.
* Training is synchronous data parallel with ONE exchange per optimiser step: every parameter's `.grad` is a view into
  one flat fp32 buffer (61.5 M parameters -> 246 MB), all-reduced in a few large buckets issued in reverse layer order
  (deepest layers' gradients are final first), then scaled by 1/world.  xGMI is point-to-point (7 links per GPU), so
  few large messages beat many small ones; BatchNorm stays per replica, like the reference's plain `nn.BatchNorm2d`
  (`models.py:43`), and rank 0's running statistics are the ones checkpointed.
"""
import os
import re

import torch
import torch.distributed as dist

# bumped by every optimiser step that changes the parameters behind torch's back (FlatAdam writes through raw pointers, which
# does not touch Tensor._version): the training engine re-packs its filter images when this or a parameter version changes
WEIGHT_EPOCH = [0]


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_indices(n_items, rank, world):
    """round-robin shard of tile indices: rank r takes r, r+world, ... (every tile exactly once, sizes differ by <= 1)"""
    return list(range(rank, n_items, world))


def shard_indices_equal(n_items, rank, world):
    """training shard: as ``shard_indices`` but every rank gets ceil(n/world) indices (the tail wraps around to the first tiles,
    like ``DistributedSampler``): ranks then run the SAME number of batches, so optimiser steps, accumulation boundaries and
    collectives line up on every rank (with shards of unequal length the ranks' ``batches_done`` drift apart and their
    all-reduces pair up across different steps, or hang at the tail)."""
    if n_items == 0:
        return []
    per = (n_items + world - 1) // world
    return [(rank + k * world) % n_items for k in range(per)]


class FlatGradReducer:
    """All parameters' gradients as views of one flat buffer + bucketed all-reduce.

    usage:  red = FlatGradReducer(model.parameters(), n_buckets=4)     # once; sets p.grad (zeros)
            loss.backward() ...                                         # autograd accumulates in place into the views
            red.all_reduce()                                            # before optimizer.step() on a step boundary
            optimizer.zero_grad(set_to_none=False)  or  red.zero()      # keep the views
    """

    def __init__(self, params, n_buckets=4, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self.offsets = []
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.offsets.append((off, n))
            off += n
        # buckets: contiguous ranges of the flat buffer, reduced from the END (deep layers) to the start
        n_buckets = max(1, min(n_buckets, len(self.params)))
        target = (total + n_buckets - 1) // n_buckets
        self.buckets, start, acc = [], 0, 0
        for (o, n) in self.offsets:
            acc += n
            if acc >= target:
                self.buckets.append((start, o + n))
                start, acc = o + n, 0
        if start < total:
            self.buckets.append((start, total))

    # ---- overlap with the backward walk --------------------------------------------------------------------------------
    def attach(self, model):
        """Let the bf16 training engine report finished layers (``model._grad_ready(layer)``, deepest layer first): a bucket's
        all-reduce is issued as soon as the gradients of its shallowest layer have been enqueued -- ``torch.distributed``
        orders the collective behind the work already on the current stream and runs it on its own stream, so it overlaps
        the backward of the layers below (between ``loss.backward()`` and ``optimizer.step()``, train.py:114-118)."""
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        assert len(names) == len(self.params)
        layer_of = [int(re.match(r"module_list\.(\d+)\.", n).group(1)) for n in names]
        self.bucket_first_layer = []
        for (a, b) in self.buckets:
            layers = [l for l, (o, n) in zip(layer_of, self.offsets) if o < b and o + n > a]
            self.bucket_first_layer.append(min(layers))
        self._armed = False
        self._handles = {}
        model._grad_ready = self.layer_done
        return self

    def begin(self):
        """arm the hooks for the next backward (the one that ends an accumulation window)"""
        self._armed = self._active()
        self._handles = {}

    def layer_done(self, layer):
        """layer < 0: the backward walk is over"""
        if not getattr(self, "_armed", False):
            return
        for k in range(len(self.buckets) - 1, -1, -1):
            if k not in self._handles and (layer < 0 or layer <= self.bucket_first_layer[k]):
                a, b = self.buckets[k]
                self._handles[k] = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or bool(os.environ.get("AY_FORCE_DIST")))

    def views_intact(self):
        return all(p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + 4 * o for p, (o, _) in zip(self.params, self.offsets))

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, average=True):
        """sum over ranks (bucketed, async, reverse order), then 1/world; returns the number of bytes exchanged per rank"""
        if not self._active():
            return 0  # (AY_FORCE_DIST: run the collectives with one rank too -- rehearsal of the N>1 path on a 1-GPU box)
        assert self.views_intact(), "parameter .grad no longer aliases the flat buffer (zero_grad(set_to_none=True)?)"
        world = dist.get_world_size(self.group)
        issued = self._handles if getattr(self, "_armed", False) else {}   # buckets already in flight (begin() + engine hooks)
        handles = [issued[k] if k in issued else dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                   for k, (a, b) in reversed(list(enumerate(self.buckets)))]
        self._armed, self._handles = False, {}
        for h in handles:
            h.wait()
        if average:
            self.flat.mul_(1.0 / world)
        return self.flat.numel() * 4


def broadcast_parameters(module, src=0, group=None):
    """make every replica start from rank `src`'s weights and BN statistics"""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not os.environ.get("AY_FORCE_DIST")):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class FlatAdam:
    """torch.optim.Adam's update (reference ``train.py:81``: default lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight
    decay) as ONE kernel over flat buffers: parameters are re-pointed to views of one flat fp32 tensor, gradients are the
    FlatGradReducer's flat buffer, moments are flat.  `grad_scale` folds the 1/world averaging into the update."""

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.red = reducer
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        flat = torch.empty_like(reducer.flat)
        for p, (o, n) in zip(reducer.params, reducer.offsets):
            flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat[o:o + n].view_as(p)
        self.flat = flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)

    def step(self, grad_scale=1.0):
        import ctypes as C
        from . import _lib
        self.step_count += 1
        WEIGHT_EPOCH[0] += 1
        _lib.check(_lib.lib().ay_adam_flat(_lib.ptr(self.flat), _lib.ptr(self.red.flat), _lib.ptr(self.m), _lib.ptr(self.v), self.flat.numel(),
                                            C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                            self.step_count, C.c_float(grad_scale), _lib.stream_ptr()), "ay_adam_flat")
        # (the packed / derived copies of the weights are stale now: the training engine re-packs when WEIGHT_EPOCH moves)
