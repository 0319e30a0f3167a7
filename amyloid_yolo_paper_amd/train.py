"""``train()``: the body of the reference's ``train.py`` (``train.py:27-210``) as a function with the same flags, plus
synchronous data parallelism when launched under ``torch.distributed.run`` (one process per GPU, RCCL).

Kept from the reference: Adam with default hyper-parameters and no schedule (``:81``); gradient accumulation that steps
when ``batches_done % gradient_accumulations == 0`` (``:116-119``, i.e. after batch 0, 2, 4, ... with gradients summed,
not averaged); per-batch metric table of the three YOLO layers (``:125-154``); ``evaluate`` at
(iou .5, conf .5, nms .5, batch 8) every ``evaluation_interval`` epochs (``:161-169``); ``state_dict`` checkpoints
``checkpoints/yolov3_ckpt_%d.pth`` (``:205-206``).  Not kept: TensorBoard logging and imgaug augmentation.
"""
import argparse
import os
import time

import torch
from torch.utils.data import DataLoader, Subset

from .datasets import ListDataset
from .models import Darknet
from .parallel import FlatAdam, FlatGradReducer, broadcast_parameters, init_distributed, shard_indices_equal
from .parse_config import parse_data_config
from .train_engine import METRIC_KEYS
from .utils import load_classes, weights_init_normal


def format_metrics(model, epoch, epochs, batch_i, n_batches):
    lines = ["---- [Epoch %d/%d, Batch %d/%d] ----" % (epoch, epochs, batch_i, n_batches)]
    lines.append("%-12s" % "Metrics" + "".join("%16s" % f"YOLO Layer {i}" for i in range(len(model.yolo_layers))))
    for k in ["grid_size"] + METRIC_KEYS[:-1]:
        fmt = "%16d" if k == "grid_size" else "%16.6f"
        lines.append("%-12s" % k + "".join(fmt % yl.metrics.get(k, 0) for yl in model.yolo_layers))
    return "\n".join(lines)


def train(epochs=100, batch_size=8, gradient_accumulations=2, model_def="config/yolov3.cfg", data_config="config/coco.data",
          pretrained_weights=None, n_cpu=8, img_size=416, checkpoint_interval=1, evaluation_interval=1,
          multiscale_training=True, verbose=False, checkpoint_dir="checkpoints", max_batches=None, seed=0, precision="bf16",
          box_loss="mse"):
    """``precision``: "bf16" = MFMA training path (bf16 activations/gradients, fp32 master weights and statistics),
    "fp32" = the parity path that reproduces the reference's fp32 step to 1e-4.  ``box_loss``: "mse" (reference) | "giou"."""
    rank, local_rank, world = init_distributed()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(seed)
    cfg = parse_data_config(data_config)
    class_names = load_classes(cfg["names"])
    model = Darknet(model_def, precision=precision).to(dev)
    model.box_loss = box_loss
    model.apply(weights_init_normal)
    if pretrained_weights:
        if pretrained_weights.endswith(".pth"):
            model.load_state_dict(torch.load(pretrained_weights))
        else:
            model.load_darknet_weights(pretrained_weights)
    broadcast_parameters(model)
    dataset = ListDataset(cfg["train"], multiscale=multiscale_training, img_size=img_size)
    # every rank gets the same number of tiles (the tail wraps around), hence the same number of batches: optimiser steps,
    # accumulation boundaries and collectives line up on all ranks
    data = Subset(dataset, shard_indices_equal(len(dataset), rank, world)) if world > 1 else dataset
    loader = DataLoader(data, batch_size=batch_size, shuffle=True, num_workers=n_cpu, pin_memory=True, collate_fn=dataset.collate_fn)
    # one exchange per optimiser step over a flat gradient buffer, overlapped with the backward on the bf16 path; the update is
    # torch.optim.Adam's (train.py:81: default hyper-parameters, no schedule) as one kernel over the flat buffers -- the step
    # bench.py --mode train times
    reducer = FlatGradReducer(model.parameters(), n_buckets=4).attach(model)
    optimizer = FlatAdam(reducer)
    ctrl = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=6))   # host-side barrier around rank 0's evaluation
    history = []
    for epoch in range(epochs):
        model.train()
        t0 = time.time()
        for batch_i, (_, imgs, targets) in enumerate(loader):
            batches_done = len(loader) * epoch + batch_i          # the same on every rank (equal shards)
            step_now = batches_done % gradient_accumulations == 0  # train.py:116: after batch 0, 2, 4, ... gradients summed
            if step_now:
                reducer.begin()                                    # buckets go out while the backward is still running
            loss, outputs = model(imgs.to(dev), targets.to(dev))
            loss.backward()
            if step_now:
                reducer.all_reduce(average=False)      # the one exchange step of data parallelism (waits for the buckets)
                optimizer.step(grad_scale=1.0 / world)
                reducer.zero()
            model.seen += imgs.size(0) * world
            history.append(float(loss.item()))
            if rank == 0 and (verbose or batch_i % 10 == 0):
                print(format_metrics(model, epoch, epochs, batch_i, len(loader)) + f"\nTotal loss {history[-1]:.4f}", flush=True)
            if max_batches is not None and batches_done + 1 >= max_batches:
                break
        if epoch % evaluation_interval == 0 and "valid" in cfg and os.path.exists(cfg["valid"]):
            if rank == 0:
                from .test import evaluate
                res = evaluate(model, cfg["valid"], 0.5, 0.5, 0.5, img_size, 8)
                if res is not None:
                    precision, recall, AP, f1, ap_class = res
                    for i, c in enumerate(ap_class):
                        print(f"+ Class '{c}' ({class_names[c]}) - AP: {AP[i]:.5f}")
                    print(f"---- mAP {AP.mean():.5f}   epoch time {time.time() - t0:.1f}s")
            if ctrl is not None:
                import torch.distributed as dist
                dist.barrier(group=ctrl)   # the other ranks wait on the host, not inside the next RCCL collective
        if rank == 0 and epoch % checkpoint_interval == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            # (parameters are views of the optimiser's flat buffer: save plain copies, the reference's .pth schema)
            torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, os.path.join(checkpoint_dir, "yolov3_ckpt_%d.pth" % epoch))
        if max_batches is not None and len(history) >= max_batches:
            break
    return model, history


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--gradient_accumulations", type=int, default=2)
    ap.add_argument("--model_def", type=str, default="config/yolov3.cfg")
    ap.add_argument("--data_config", type=str, default="config/coco.data")
    ap.add_argument("--pretrained_weights", type=str)
    ap.add_argument("--n_cpu", type=int, default=8)
    ap.add_argument("--img_size", type=int, default=416)
    ap.add_argument("--checkpoint_interval", type=int, default=1)
    ap.add_argument("--evaluation_interval", type=int, default=1)
    ap.add_argument("--compute_map", default=False)
    ap.add_argument("--multiscale_training", default=True)
    ap.add_argument("--verbose", "-v", default=False, action="store_true")
    ap.add_argument("--logdir", type=str, default="logs")
    ap.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--box_loss", type=str, default="mse", choices=["mse", "giou"])
    o = ap.parse_args(argv)
    train(o.epochs, o.batch_size, o.gradient_accumulations, o.model_def, o.data_config, o.pretrained_weights, o.n_cpu, o.img_size,
          o.checkpoint_interval, o.evaluation_interval, o.multiscale_training not in (False, "False"), o.verbose,
          precision=o.precision, box_loss=o.box_loss)


if __name__ == "__main__":
    main()
