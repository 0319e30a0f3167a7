#!/usr/bin/env python3
"""Headline benchmark: tiles/s of the YOLOv3 hot path (1024x1024 RGB tiles, bf16 MFMA backbone+FPN, decode, merge-NMS)
on MI355X.  One "step" = one pass over one batch of 64 device-resident synthetic tiles (BASELINE.json configs[1]).

    python bench.py [--gpus N --steps K --warmup W]

N > 1 runs one rank per GPU either way it is started: under `python -m torch.distributed.run --nproc-per-node N` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* come from the environment), or from a plain `python bench.py --gpus N`, in which case this process starts the
N ranks itself as child processes -- before it imports torch or touches HIP -- waits for them, forwards rank 0's JSON line and
exits with their return code (`launch_ranks`).

Prints ONE JSON line on rank 0 (contract in the task description), with `roofline` for the dominant kernel
(3x3 stride-1 MFMA convolution, BN=128 tile) timed with HIP events on the launch stream inside the timed region and
`cpu_baseline` = the CPU oracle (torch-CPU fp32 + restated decode/NMS) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

USE_DIST = False
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_F32_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU x 256 CUs x 2.4 GHz (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: tiles/s of the inference hot path (headline); train: imgs/s of the training step (configs[2]/[3])")
    ap.add_argument("--train_batch", type=int, default=32)
    ap.add_argument("--train_size", type=int, default=416, help="training tile side (reference default train.py:36)")
    ap.add_argument("--no_other_dtype", action="store_true", help="default mode: skip the timing of the other 16-bit storage type")
    ap.add_argument("--no_train_leg", action="store_true", help="default mode: skip the appended training measurement (configs[2])")
    ap.add_argument("--no_fp32_leg", action="store_true", help="default mode: skip the appended fp32 parity-path measurement")
    ap.add_argument("--no_pipelined_leg", action="store_true", help="default mode: skip the appended two-batches-in-flight measurement")
    ap.add_argument("--leg_train_size", type=int, default=1024, help="tile side of the appended training measurement (configs[2])")
    ap.add_argument("--leg_train_steps", type=int, default=10)
    ap.add_argument("--fp32_batch", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16",
                    help="16-bit storage type of the timed inference path: bf16 = BASELINE.json configs[1] as written (and the faster one: the "
                         "power-limited chip clocks 4 %% lower on fp16 operands); fp16 = the more faithful one (0.98 vs 0.83 of the reference's NMS "
                         "indices).  The other type's timing and parity ride in the same line (other_dtype, parity_<other>); DESIGN.md section 2")
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--conf_thres", type=float, default=0.5)
    ap.add_argument("--nms_thres", type=float, default=0.4)
    ap.add_argument("--max_det", type=int, default=2048)
    ap.add_argument("--unique_tiles", type=int, default=16, help="distinct synthetic tiles (repeated to fill the batch)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_tiles", type=int, default=16, help="tiles in the CPU-oracle sample (configs[0]: 16 tiles, ~12 s)")
    ap.add_argument("--serial_nms", action="store_true", help="run merge-NMS on the main stream (no overlap with the next batch)")
    ap.add_argument("--no_layer_events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--event_every", type=int, default=4,
                    help="HIP event pairs around the dominant family's launches on every n-th step of the timed region (the first included): "
                         "62 event records cost the 24-ms step 0.17 ms when every step carries them (same-box A/B)")
    ap.add_argument("--launch_probe", action="store_true",
                    help="rank plumbing only (no GPU): every rank joins the gloo rendezvous the timed run uses, the ranks' numbers are summed "
                         "and rank 0 prints one JSON line; tests/test_dp_cpu.py drives the self-launcher through this on CPU-only boxes")
    ap.add_argument("--traffic_json", default=os.path.join(REPO, "profiles", "traffic.json"),
                    help="optional {kernel family: HBM bytes per launch} from a rocprofv3 --pmc pass")
    return ap.parse_args()


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources: profiles/traffic.json records the value it was measured on, and a
    traffic figure from other sources is not reported (the GPU box has no .git to ask for a commit)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "amyloid_yolo_paper_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def conv_flops(e, B, S):
    h = S >> e["log2_down"]
    return 2.0 * B * h * h * e["cout"] * e["cin"] * e["k"] * e["k"]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process (which has not imported torch and
    never touches the GPU: nothing that has initialised HIP is exec'd or forked), one per GPU, rendezvous on 127.0.0.1.  Rank 0's
    stdout (the ONE JSON line) is forwarded as it is; the other ranks' stdout goes to stderr.  Return code: 0 when every rank
    returned 0, else the first non-zero one; when a rank fails the others are terminated (their own PIDs), nothing is retried."""
    import signal
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": os.environ.get("MASTER_PORT") or str(free_port())})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = set(range(n))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    for o in alive:
                        procs[o].terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for r in alive:
            procs[r].send_signal(signal.SIGINT)
        rc = 130
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        a.gpus = world          # the launcher's world size wins (torch.distributed.run --nproc-per-node N)
    if a.launch_probe:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank), float(local_rank), 1.0], dtype=torch.float64)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"probe": "launch", "n_gpus": a.gpus, "world": world, "sum_ranks": t[0].item(), "sum_local_ranks": t[1].item(),
                              "ranks_seen": int(t[2].item()), "master": os.environ["MASTER_ADDR"]}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py (rank {rank} of {world}) needs a HIP device: no CPU fallback for the product path")
    if os.environ.get("AY_BENCH_SHARE_GPU"):   # rehearsal of the N-rank launch on a box with fewer GPUs than ranks: ranks share devices
        local_rank %= torch.cuda.device_count()  # (gloo legs only: RCCL refuses two ranks on one device, so pass --no_train_leg)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # AY_FORCE_DIST=1 walks the process-group code path with a single rank too (rehearsal of the N>1 launch on a 1-GPU box)
    global USE_DIST
    USE_DIST = world > 1 or bool(os.environ.get("AY_FORCE_DIST"))
    if USE_DIST:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # inference replicas exchange nothing on the data path: rendezvous, barriers and the max-over-ranks of the elapsed time
        # go over gloo (host).  With an RCCL communicator merely initialised the same step measured 0.8 ms (3 %) slower
        # (same-box A/B), so RCCL is brought up only where it is used: the gradient all-reduce of --mode train.
        if a.mode == "train":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from amyloid_yolo_paper_amd import _lib, build as aybuild, cfg_gen, parse_config, synth
    from amyloid_yolo_paper_amd.models import Darknet
    from amyloid_yolo_paper_amd.utils import nms_device

    if a.mode == "train":
        return bench_train(a, rank, local_rank, world, dev)

    if not os.path.exists(_lib.LIB_PATH):
        if rank == 0:
            aybuild.build_library(verbose=False)
        if USE_DIST:
            dist.barrier()

    cfg = cfg_gen.write_cfg(a.classes)
    defs = parse_config.parse_model_config(cfg)
    params = synth.synth_params(defs, seed=7)
    def make_model(precision):
        m = Darknet(cfg, img_size=a.size, precision=precision)
        sd = m.state_dict()
        for i, p in params.items():
            for k, name in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"),
                            ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
                if k in p:
                    sd[f"module_list.{i}.{name}"].copy_(torch.from_numpy(p[k]))
        return m.to(dev).eval()

    model = make_model(a.dtype)

    # synthetic tiles, resident in HBM before the timed region (each rank its own shard of tile indices: weak scaling)
    nu = min(a.unique_tiles, a.batch)
    tiles = synth.synth_tiles(nu, a.size, start=rank * nu)
    x = torch.from_numpy(tiles).to(dev)
    x = x.repeat((a.batch + nu - 1) // nu, 1, 1, 1)[: a.batch].contiguous()

    # dominant kernel family: 3x3 stride-1 convs whose padded cout is a multiple of 128
    fam = [i for i, e in enumerate(model._graph) if e["type"] == "convolutional" and e["k"] == 3 and e["stride"] == 1
           and e["cout"] % 128 == 0 and e["cin"] % 16 == 0 and not (model.fuse_blocks and e.get("in_fused_block"))]
    fam_flops = sum(conv_flops(model._graph[i], a.batch, a.size) for i in fam)
    total_flops = sum(conv_flops(e, a.batch, a.size) for e in model._graph if e["type"] == "convolutional")

    # The decode output is double-buffered and merge-NMS of batch i runs on a side stream while the convolutions of
    # batch i+1 run on the main stream (NMS keeps 64 of 256 CUs busy with one wave each; it hides completely).
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    slot_free = [None, None]
    counter = [0]

    def step():
        slot = counter[0] & 1
        counter[0] += 1
        if slot_free[slot] is not None:
            main.wait_event(slot_free[slot])          # NMS that read this output slot two steps ago is done
        out = model.forward_device(x, out_slot=slot)
        if a.serial_nms:
            return nms_device(out, a.conf_thres, a.nms_thres, a.max_det, slot)
        ready = torch.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            res = nms_device(out, a.conf_thres, a.nms_thres, a.max_det, slot)
            done = torch.cuda.Event()
            done.record(side)
        slot_free[slot] = done
        return res

    for _ in range(a.warmup):
        res = step()
    torch.cuda.synchronize()
    if USE_DIST:
        dist.barrier()
    if not a.no_layer_events:
        if model.use_plan:  # HIP event pairs around the family's launches, recorded by the native plan on the stream it issues to
            model.plan_profile_begin(a.batch, a.size, set(fam), every=a.event_every)
        else:               # AY_USE_PLAN=0: the per-layer walk brackets the same launches itself
            model.profile_layers = set(fam)
            model.profile_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    torch.cuda.synchronize()
    if USE_DIST:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if USE_DIST:
        t = torch.tensor([elapsed], dtype=torch.float64)  # host tensor: the inference group is gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    events = []
    if not a.no_layer_events and model.use_plan:
        per_op, n_fwd = model.plan_profile_end(a.batch, a.size)
        assert n_fwd == (a.steps + a.event_every - 1) // a.event_every   # the forwards that carried event pairs
        famset = set(fam)
        events = [ms for layer, kind, ms in per_op if layer in famset]   # per family launch: ms summed over the timed steps
    elif not a.no_layer_events:
        n_fwd = a.steps
        by_layer = {}
        for layer, s, e in model.profile_events:
            by_layer[layer] = by_layer.get(layer, 0.0) + s.elapsed_time(e)
        events = list(by_layer.values())
        model.profile_layers = None

    rows, keep, count, cand = res
    cnt, cnd = count.cpu().numpy(), cand.cpu().numpy()
    # the other 16-bit storage type on the same workload, same protocol (own warm-up, barrier + synchronize around the same number
    # of steps): reported beside the headline, never in place of it
    other = "bf16" if a.dtype == "fp16" else "fp16"
    other_line = None
    if not a.no_other_dtype:
        main_model, model = model, make_model(other)
        slot_free[0] = slot_free[1] = None
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        if USE_DIST:
            dist.barrier()
        torch.cuda.synchronize()
        to0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        if USE_DIST:
            dist.barrier()
        el_o = time.perf_counter() - to0
        if USE_DIST:
            t = torch.tensor([el_o], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_o = float(t.item())
        other_line = {"dtype": other, "value": round(a.gpus * a.batch * a.steps / el_o, 2), "unit": "tiles/s", "ms_per_step": round(1e3 * el_o / a.steps, 3)}
        other_model, model = model, main_model
    # The same step with TWO batches in flight: two model instances (own arenas and output slots) on two streams, steps alternating
    # between them, each stream running its forward and its merge-NMS back to back.  A persistent launch takes every CU, so nothing of
    # one batch hides under the other except the tail of a launch (workgroups that have run out of items), the ramp of the next one and
    # the gaps in between -- which is what this leg measures.  It is reported beside the headline, never as `value`: once launches of
    # two streams overlap, the per-launch HIP-event times the roofline is built from include the other stream's work.
    pipelined = None
    if not a.no_pipelined_leg and a.gpus == 1:
        try:
            pipes = [(model, torch.cuda.Stream(device=dev), 0), (make_model(a.dtype), torch.cuda.Stream(device=dev), 2)]
            torch.cuda.synchronize()

            def pstep(k):
                m_, st_, slot_ = pipes[k & 1]
                with torch.cuda.stream(st_):
                    o_ = m_.forward_device(x, out_slot=0)
                    return nms_device(o_, a.conf_thres, a.nms_thres, a.max_det, slot_)

            for k in range(2 * max(1, a.warmup // 2)):
                pres = pstep(k)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for k in range(a.steps):
                pres = pstep(k)
            torch.cuda.synchronize()
            el_p = time.perf_counter() - tp0
            same = bool(torch.equal(pres[2], count))
            pipelined = {"value": round(a.batch * a.steps / el_p, 2), "unit": "tiles/s", "ms_per_step": round(1e3 * el_p / a.steps, 3), "dtype": a.dtype,
                         "batches_in_flight": 2, "same_detection_counts_as_the_headline_step": same,
                         "what": "two batches in flight on two streams (two model instances, forward + merge-NMS back to back on each): the launch tails, "
                                 "ramps and gaps of one batch filled by the other; not the headline because overlapping launches void per-launch event times"}
            del pipes, pres
        except Exception as exc:
            pipelined = {"error": f"{type(exc).__name__}: {exc}"}
    assert cnt.max() <= a.max_det, f"max_det {a.max_det} too small: {cnt.max()} cluster heads"
    raw_ok = bool(torch.isfinite(rows[0, : max(int(cnt[0]), 1)]).all())
    assert raw_ok, "non-finite detections"

    tiles_per_s = a.gpus * a.batch * a.steps / elapsed
    result = {
        "metric": "tiles/sec (1024x1024 RGB) inference", "value": round(tiles_per_s, 2), "unit": "tiles/s",
        "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"configs[1]: Darknet-53+FPN YOLOv3, batch={a.batch} synthetic {a.size}x{a.size} tiles/GPU, "
                               f"C={a.classes}, seeded random weights, decode + merge-NMS on (conf {a.conf_thres}, nms {a.nms_thres})",
                   "global_batch": a.gpus * a.batch, "tile": a.size, "parallelism": f"replicas x{a.gpus} (tiles sharded by image, no collective)",
                   "candidates_per_tile": round(float(cnd.mean()), 1), "detections_per_tile": round(float(cnt.mean()), 1),
                   "model_tflops": round(total_flops * a.steps * a.gpus / elapsed / 1e12, 1)},
    }
    if other_line is not None:
        result["other_dtype"] = other_line
    if pipelined is not None:
        result["pipelined"] = pipelined
    if rank == 0:
        if events:
            ms = sum(events)
            launches = len(events) * n_fwd                      # launches that were bracketed by an event pair
            achieved = fam_flops * n_fwd / (ms * 1e-3) / 1e12
            traffic, traffic_note = load_traffic(a, os.path.basename(a.traffic_json), "conv3x3s1_bn128_bytes_per_launch")
            result["roofline"] = {
                "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                "kernel": f"ay::conv3x3_m16_ring_kernel<res|nores, {'F16' if a.dtype == 'fp16' else 'Bf16'}> (3x3 s1, 128 ch x 512 px tile, "
                          f"v_mfma_f32_16x16x32_{'f16' if a.dtype == 'fp16' else 'bf16'}, persistent LDS-DMA ring; "
                          "AY_M16=0: ay::conv_bf16_ring_kernel<3,1,128,2,4,16,32,1,2,...> on 32x32x16)",
                "launches_per_step": launches // n_fwd, "avg_launch_ms": round(ms / launches, 4),
                "flops_per_launch": fam_flops / (launches // n_fwd), "family_share_of_model_flops": round(fam_flops / total_flops, 3),
                "timed_launches": launches, "launch_sample": f"HIP event pairs around every launch of the family on {n_fwd} of the {a.steps} timed steps "
                                                             f"(every {a.event_every}th, inside the timed region, on the issue stream)",
            }
    # ---- the rest of BASELINE.json's metric in the same line: "train imgs/sec @1/2/4/8 GPU" (configs[2]: B=32 per GPU, 1024^2), and the
    # parity-grade fp32 path's tiles/s.  All GPU legs first, the host-heavy CPU baseline + parity legs last.  RCCL comes up only now
    # (an initialised communicator costs the inference step 3 %, see above) as a second process group next to the gloo one.
    del x, res, rows, keep, count, cand
    if not a.no_train_leg:
        group = None
        try:
            if USE_DIST:
                import datetime
                # a rank that fails alone (e.g. out of memory) must not leave the others waiting in a collective for ever
                group = dist.new_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
            tr = measure_train(a, rank, world, dev, a.train_batch, a.leg_train_size, a.leg_train_steps, 2, group=group)
            result["train"] = {k: tr[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "config",
                                                    "all_reduce_bytes", "all_reduce_buckets", "roofline") if k in tr}
        except Exception as exc:  # the headline number stands on its own
            result["train"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        if not a.no_fp32_leg and a.gpus == 1:
            try:
                result["fp32_path"] = measure_fp32_path(a, cfg, params, dev)
            except Exception as exc:
                result["fp32_path"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not a.no_cpu_baseline and a.gpus == 1:
            result["cpu_baseline"], ref_dets, ref_tiles = cpu_baseline(a, cfg, params)
            result["parity"] = parity_vs_cpu(a, model, ref_dets, ref_tiles)     # the dtype that was timed
            result["parity_" + other] = parity_vs_cpu(a, other_model if other_line is not None else make_model(other), ref_dets, ref_tiles)  # the other 16-bit storage type, same tiles
        print(json.dumps(result), flush=True)
    if USE_DIST:
        dist.barrier()
        dist.destroy_process_group()


def measure_train(a, rank, world, dev, B, S, steps, warmup, group=None):
    """BASELINE.json configs[2]/[3]: random-init YOLOv3 (3 classes), batch B per GPU, synthetic boxes; one step = forward
    (train-mode BN, loss) + backward + gradient all-reduce over RCCL (N > 1, `group`) + Adam, all inside the timed region.
    Barrier + synchronize on both sides, max over ranks.  Returns the result dict (every rank)."""
    import torch
    import torch.distributed as dist
    from amyloid_yolo_paper_amd import cfg_gen, synth
    from amyloid_yolo_paper_amd.models import Darknet
    from amyloid_yolo_paper_amd.parallel import FlatAdam, FlatGradReducer, broadcast_parameters
    from amyloid_yolo_paper_amd.utils import weights_init_normal
    torch.manual_seed(1234)
    model = Darknet(cfg_gen.write_cfg(a.classes), img_size=S, precision="bf16").to(dev)
    model.apply(weights_init_normal)
    broadcast_parameters(model, group=group)
    model.train()
    nu = min(8, B)
    x = torch.from_numpy(synth.synth_tiles(nu, S, start=100 + rank * nu)).to(dev)
    x = x.repeat((B + nu - 1) // nu, 1, 1, 1)[:B].contiguous()
    tg = torch.from_numpy(synth.synth_targets(B, a.classes, seed=77 + rank, grid=S // 8)).to(dev)
    red = FlatGradReducer(model.parameters(), n_buckets=4, group=group).attach(model)
    opt = FlatAdam(red)
    losses = []
    model.collect_metrics = False   # the per-layer metric table (one host sync per step) is train()'s logging, not the step

    def step():
        red.begin()                 # N > 1: each bucket's all-reduce is issued from inside the backward walk
        loss, _ = model.train_step_device(x, tg)
        loss.backward()
        red.all_reduce(average=False)
        opt.step(grad_scale=1.0 / world)
        red.zero()
        return loss

    def barrier():
        if USE_DIST:
            dist.barrier()          # default group: gloo in the default mode (host), RCCL under --mode train

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    prof_acc = None if a.no_layer_events else {"wgrad": [], "conv": []}   # HIP event pairs around the 3x3 family's launches (issue stream),
    n_prof = 0                                                            # on every a.event_every-th step of the timed region
    t0 = time.perf_counter()
    for k in range(steps):
        on = prof_acc is not None and k % a.event_every == 0
        model._train_prof = prof_acc if on else None
        n_prof += int(on)
        losses.append(step())
    model._train_prof = prof_acc
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if USE_DIST:
        on_host = dist.get_backend() == "gloo"
        t = torch.tensor([elapsed], device="cpu" if on_host else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lv = [float(l.item()) for l in losses]
    assert all(v == v and abs(v) != float("inf") for v in lv), "non-finite training loss"
    ar_bytes = red.flat.numel() * 4 if red._active() else 0     # fp32 gradient bytes each rank contributes to the exchange, per step
    n_buckets = len(red.buckets)
    flops_per_img = 3.0 * sum(conv_flops(e, 1, 1024) for e in model._graph if e["type"] == "convolutional") * (S / 1024.0) ** 2
    ctx_bytes = sum(c.bytes() for c in getattr(model, "_train_ctx", {}).values())
    result = {
        "metric": "train imgs/sec", "value": round(world * B * steps / elapsed, 2), "unit": "imgs/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"configs[2]: random-init YOLOv3 C={a.classes}, batch={B}/GPU, {S}x{S} synthetic tiles + boxes, train-mode BN, "
                               f"loss + backward + {'RCCL all-reduce (246 MB fp32, 4 buckets, issued from inside the backward) + ' if world > 1 else ''}Adam",
                   "global_batch": world * B, "tile": S, "parallelism": f"dp{world}", "first_loss": round(lv[0], 3), "last_loss": round(lv[-1], 3),
                   "model_tflops": round(world * B * steps * flops_per_img / elapsed / 1e12, 1),
                   "hbm_peak_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1), "saved_activations_gb": round(ctx_bytes / 1e9, 1)},
        "all_reduce_bytes": ar_bytes, "all_reduce_buckets": n_buckets if ar_bytes else 0,
    }
    prof = getattr(model, "_train_prof", None)
    if prof and prof["wgrad"]:
        fam = [e for e in model._graph if e["type"] == "convolutional" and e["k"] == 3 and e["stride"] == 1 and e["cout"] % 128 == 0 and e["cin"] % 32 == 0]
        fl_w = sum(conv_flops(e, B, S) for e in fam)                                        # weight gradients of the family, per step
        fl_c = fl_w + sum(conv_flops(e, B, S) for e in fam if e["cin"] % 128 == 0)           # forward + the timed data gradients
        ms_w = sum(e0.elapsed_time(e1) for e0, e1 in prof["wgrad"])
        ms_c = sum(e0.elapsed_time(e1) for e0, e1 in prof["conv"])
        ach_w = fl_w * n_prof / (ms_w * 1e-3) / 1e12
        traffic, traffic_note = load_traffic(a, "traffic_train.json", "wgrad3x3_bytes_per_launch") if (B, S) == (32, 1024) else (None, "measured at B=32, 1024^2 only")
        result["roofline"] = {
            "bound": "mfma", "achieved": round(ach_w, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach_w / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_note,
            "kernel": "ay::wgrad_bf16_kernel<3,1> (weight gradient of the 3x3 s1 family: the largest share of the step)",
            "launches_per_step": len(prof["wgrad"]) // n_prof, "avg_launch_ms": round(ms_w / len(prof["wgrad"]), 4),
            "flops_per_launch": fl_w / max(1, len(prof["wgrad"]) // n_prof),
            "share_of_step": round(ms_w / n_prof / (1e3 * elapsed / steps), 3),
            "launch_sample": f"HIP event pairs on {n_prof} of the {steps} timed steps (every {a.event_every}th)",
            "conv_family": {"kernel": "ay::conv3x3_m16_ring_kernel (forward + data gradient of the same layers; grids below 16 rows or canvas-tiled ones -- the 416-px maps -- run on ay::conv_bf16_ring_kernel<3,1,128,...>)",
                            "achieved": round(fl_c * n_prof / (ms_c * 1e-3) / 1e12, 1), "frac": round(fl_c * n_prof / (ms_c * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                            "launches_per_step": len(prof["conv"]) // n_prof, "share_of_step": round(ms_c / n_prof / (1e3 * elapsed / steps), 3)},
        }
    # hand the step's memory back: the default mode goes on to other legs
    model._train_ctx = {}
    del model, red, opt, x, tg, losses
    torch.cuda.empty_cache()
    return result


def bench_train(a, rank, local_rank, world, dev):
    import torch.distributed as dist
    result = measure_train(a, rank, world, dev, a.train_batch, a.train_size, a.steps, a.warmup)
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_train_baseline(a)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if USE_DIST:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(a, fname, key):
    """HBM bytes per launch of a kernel family from a committed rocprofv3 --pmc summary under profiles/, reported only when the
    file was measured on THESE kernel sources (csrc_sha16; the GPU box has no .git to ask for a commit)."""
    path = os.path.join(os.path.dirname(a.traffic_json), fname)
    try:
        tj = json.load(open(path))
    except Exception:
        return None, f"no profiles/{fname}"
    if tj.get("csrc_sha16") != csrc_sha16():
        return None, f"profiles/{fname} was measured on other kernel sources (csrc_sha16 differs): not reported"
    return tj.get(key), tj.get("source")


def measure_fp32_path(a, cfg, params, dev):
    """The parity-grade path (precision="fp32": the one that meets north_star's bit-exact-indices / 1e-4 clause end to end) on the
    same workload at batch --fp32_batch: tiles/s incl. decode + merge-NMS, with the roofline of its convolution stack against the
    exact-fp32 MFMA peak."""
    import torch
    from amyloid_yolo_paper_amd.models import Darknet
    from amyloid_yolo_paper_amd.utils import nms_device
    from amyloid_yolo_paper_amd import synth
    B, S = a.fp32_batch, a.size
    m = Darknet(cfg, img_size=S, precision="fp32")
    sd = m.state_dict()
    for i, p in params.items():
        for k, name in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"),
                        ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
            if k in p:
                sd[f"module_list.{i}.{name}"].copy_(torch.from_numpy(p[k]))
    m = m.to(dev).eval()
    x = torch.from_numpy(synth.synth_tiles(B, S, start=0)).to(dev)

    def step():
        return nms_device(m.forward_device(x), a.conf_thres, a.nms_thres, a.max_det, 0)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    steps = max(2, min(20, int(4.0 / max(one, 1e-3))))   # about 4 s of it
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev0.record()
        out = m.forward_device(x)
        ev1.record()
        res = nms_device(out, a.conf_thres, a.nms_thres, a.max_det, 0)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    total_flops = sum(conv_flops(e, B, S) for e in m._graph if e["type"] == "convolutional")
    fwd_ms = ev0.elapsed_time(ev1)   # the last step's convolution stack + decode (events on the issue stream)
    ach = total_flops / (fwd_ms * 1e-3) / 1e12
    return {"value": round(B * steps / elapsed, 2), "unit": "tiles/s", "dtype": "f32", "batch": B, "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 2),
            "detections_per_tile": round(float(res[2].float().mean().item()), 1),
            "what": "precision='fp32' (NCHW fp32 activations and filters, fp32 accumulate): the path that is bit-exact in NMS indices and within 1e-4 in "
                    "boxes against the reference on all five model fixtures (tests/test_gpu_parity.py::test_model_fp32_vs_reference_fixtures)",
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_TFLOPS, 4),
                         "kernel": "ay::conv_f32_mfma_kernel<KS,STRIDE,KC,DUAL> (v_mfma_f32_32x32x2_f32: exact fp32 products and sums; 64 ch x 8x32 px per "
                                   "workgroup, NCHW fp32 / OIHW fp32 as the reference lays them out; csrc/ay_conv_f32_mfma.hip)",
                         "what": "whole convolution stack of the fp32 path (395.65 GFLOP per 1024^2 tile) over its HIP-event time, against the exact-fp32 MFMA peak"}}


def cpu_train_baseline(a):
    """BASELINE.md section 3, training leg: the CPU oracle (fp32 torch-CPU autograd through the restated graph and loss, pinned to
    the reference's training step by tests/golden/train_*.npz), B=2 at 416^2, 1 warm-up + 4 timed steps with torch.optim.Adam."""
    import torch
    from amyloid_yolo_paper_amd import cfg_gen, synth
    from oracle.darknet_oracle import OracleDarknet
    cores = host_cores()
    torch.set_num_threads(cores)
    B, S = 2, 416
    m = OracleDarknet(cfg_gen.write_cfg(a.classes))
    g = torch.Generator().manual_seed(1234)
    for p in m.params.values():
        p["weight"] = torch.randn(p["weight"].shape, generator=g) * 0.02
        if "gamma" in p:
            p["gamma"] = 1.0 + 0.02 * torch.randn(p["gamma"].shape, generator=g)
    m.require_grad()
    params = [t for p in m.params.values() for k, t in p.items() if k in ("weight", "gamma", "beta", "bias")]
    opt = torch.optim.Adam(params)
    x = torch.from_numpy(synth.synth_tiles(B, S, start=100))
    tg = torch.from_numpy(synth.synth_targets(B, a.classes, seed=77, grid=S // 8))
    times = []
    for it in range(5):
        t0 = time.perf_counter()
        loss, _ = m.forward(x, tg, train_bn=True)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:])
    return {"value": round(B * 4 / dt, 3), "unit": "imgs/s", "cores": cores, "kind": "port",
            "sample": f"4 timed steps (after 1 warm-up) of batch {B} at {S}x{S}: forward (train-mode BN) + loss + backward + Adam, fp32 torch-CPU "
                      f"autograd through the restated graph on {cores} threads ({dt:.1f} s); the GPU line is batch {a.train_batch} at {a.train_size}^2"}


def host_cores():
    """threads the CPU baseline may use: scheduler affinity, clipped by the cgroup CPU quota of this box"""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(cores, int(os.environ.get("AY_CPU_THREADS", "16"))))  # GPU boxes give 16 CPUs per GPU


def cpu_baseline(a, cfg, params):
    """The CPU oracle (port of the reference's CPU path, pinned to it by tests/golden) on a bounded sample of the same
    workload, on the host cores of this box, by the protocol of BASELINE.md section 3 (which mirrors validation.speedCheck:
    data loading excluded): --cpu_tiles synthetic tiles at batch 1 in >= 3 timed passes after a warm-up, then the same tiles
    at batch 8.  Returns (json dict, per-tile reference detections, the tiles)."""
    import torch
    from amyloid_yolo_paper_amd import synth
    from oracle import boxes_oracle as bo
    from oracle.darknet_oracle import OracleDarknet
    cores = host_cores()
    torch.set_num_threads(cores)
    m = OracleDarknet(cfg)
    m.set_params(params)
    tiles = torch.from_numpy(synth.synth_tiles(a.cpu_tiles, a.size, start=0))
    dets = []
    with torch.no_grad():
        tw = time.perf_counter()
        m.forward(tiles[:1])  # warm-up, also sizes the sample
        warm = time.perf_counter() - tw
        n = max(4, min(a.cpu_tiles, int(24.0 / max(warm, 1e-3)))) if a.cpu_tiles >= 4 else a.cpu_tiles
        n_pass = 4 if n >= 4 else 1
        per = n // n_pass
        n = per * n_pass
        pass_rates, t_model, t_nms = [], 0.0, 0.0
        for p in range(n_pass):
            t0 = time.perf_counter()
            for i in range(p * per, (p + 1) * per):
                t1 = time.perf_counter()
                out = m.forward(tiles[i:i + 1]).numpy()
                t2 = time.perf_counter()
                rows, keep, _ = bo.non_max_suppression(out, a.conf_thres, a.nms_thres)
                t3 = time.perf_counter()
                t_model += t2 - t1
                t_nms += t3 - t2
                dets.append((keep[0], rows[0]))
            pass_rates.append(per / (time.perf_counter() - t0))
        b1 = n / (t_model + t_nms)
        # batch 8 over the same tiles (at most 16 s of it)
        b8, nb8 = None, 0
        if n >= 8:
            t0 = time.perf_counter()
            for i in range(0, n - 7, 8):
                out = m.forward(tiles[i:i + 8]).numpy()
                bo.non_max_suppression(out, a.conf_thres, a.nms_thres)
                nb8 += 8
                if time.perf_counter() - t0 > 16.0:
                    break
            b8 = nb8 / (time.perf_counter() - t0)
    best = max(b1, b8 or 0.0)
    res = {"value": round(best, 4), "unit": "tiles/s", "cores": cores, "kind": "port",
           "batch1_tiles_per_s": round(b1, 4), "batch8_tiles_per_s": None if b8 is None else round(b8, 4),
           "batch1_passes_tiles_per_s": [round(r, 4) for r in pass_rates],
           "sample": f"{n} of the same synthetic {a.size}x{a.size} tiles: batch 1 in {n_pass} timed passes of {per} after a warm-up "
                     f"(model {t_model:.1f} s, decode+merge-NMS {t_nms:.2f} s), then {nb8} of them at batch 8; fp32 torch-CPU conv stack "
                     f"+ restated decode/merge-NMS on {cores} threads; value = the faster of the two batch sizes"}
    return res, dets, tiles[:n]


def parity_vs_cpu(a, model, ref_dets, tiles):
    """The timed (bf16) path against the CPU oracle's detections on the cpu_baseline sample: share of the oracle's kept box
    indices it reproduces, largest box / confidence difference over the matched heads (oracle/parity.py).  The bars this is held
    to, and why a path that stores bf16 activations cannot match every index, are in tests/test_gpu_configs.py."""
    import torch
    from amyloid_yolo_paper_amd.utils import non_max_suppression
    from oracle import parity
    items = []
    with torch.no_grad():
        for i, (keep, rows) in enumerate(ref_dets):
            out = model.forward_device(tiles[i:i + 1]).clone()
            res = non_max_suppression(out, a.conf_thres, a.nms_thres)
            got = None if res[0] is None else res[0].cpu().numpy()
            items.append(parity.detection_agreement(keep, rows, res.keep_idx[0], got))
    return dict(parity.summarize(items), dtype=model.precision, against="CPU oracle (fp32, = the reference's CPU path on tests/golden), same tiles, "
                                                  "conf %.2f nms %.2f" % (a.conf_thres, a.nms_thres))


if __name__ == "__main__":
    main()
