"""BASELINE.json configurations at their full shape on the GPU (-m gpu).

configs[1]: bf16 MFMA path, batch 64 of 1024x1024 tiles, decode + merge-NMS -- the configuration bench.py times -- against the
reference's own output for the 1024^2 fixture tile (tests/golden/model_c3_s1024_b1.npz, produced by oracle/gen_golden.py
from the imported reference: models.py:237-255 + utils/utils.py:235-273)."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from amyloid_yolo_paper_amd import utils as ay
from oracle import parity
from test_gpu_parity import build_models, load

pytestmark = pytest.mark.gpu

# Bars for the bf16 product path against the reference's fp32 CPU result.  north_star asks for bit-exact indices and 1e-4
# boxes "on identical tiles"; the fp32 HIP path meets exactly that (test_model_fp32_vs_reference_fixtures, 1024^2 included).  A
# path that stores every activation in bf16 cannot: a confidence within bf16 noise (~0.02) of the 0.5 threshold enters or
# leaves the candidate set, and the synthetic heads are calibrated so that the threshold sits in the bulk of the
# distribution (on the fixture tile 521 of 64 512 rows lie within 0.02 of it, against 445 candidates).  What the rounding
# CONTRACT itself attains was measured with the CPU oracle in mode="bf16" on this tile (exact same rounding points, fp32 and
# float64 convolution sums): 0.817 / 0.835 of the reference's 230 kept indices, 0.126 / 0.090 extra heads, confidences of
# matched heads within 0.023, boxes of matched heads within 0.33 of the box size (a head whose cluster gained or lost a
# member moves with the conf-weighted merge, utils/utils.py:259-269).  The HIP path is held to the contract's own level:
# Round 3: the same contract on IEEE-half storage (precision="fp16", an 11-bit significand against bf16's 8: every rounding step 8x
# smaller, same bytes, same MFMA rate) attains, by the CPU oracle in mode="fp16" on this tile (fp32 / float64 sums): 0.9957 / 1.0 of
# the 230 kept indices, 0.017 / 0.021 extra heads, matched confidences within 0.0027, matched boxes within 0.168 of the box size
# (one cluster gains a member).  The half path (bench.py --dtype fp16, `other_dtype` / `parity_fp16` of every default line) is held to that level.
BARS = {   # keep_match >=, extra_heads <=, max_box_rel <=, max_dconf <=, decoded-sample dconf q99 <=, box rel q99.9 <=
    "bf16": (0.78, 0.16, 0.40, 0.03, 2e-2, 5e-2),
    "fp16": (0.98, 0.035, 0.20, 0.004, 2.5e-3, 6.25e-3),
}


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_configs1_bf16_b64_1024(golden_dir, tmp_cfg_dir, precision):
    KEEP_MATCH_MIN, EXTRA_HEADS_MAX, BOX_REL_MAX, DCONF_MAX, DCONF_Q99, REL_Q999 = BARS[precision]
    dev = torch.device("cuda", 0)
    name, C_, S, B1, start = [c for c in gc.MODEL_CASES if c[0] == "c3_s1024_b1"][0]
    z = load(golden_dir, "model_" + name)
    m, _ = build_models(C_, tmp_cfg_dir, dev, precision)
    assert m.use_plan and not m.keep_layer_outputs
    x1 = torch.from_numpy(gc.model_inputs(S, B1, start))            # the fixture's tile
    out1 = m.forward_device(x1).clone()
    assert bool(torch.isfinite(out1).all())
    # the same tile 64 times through the native plan (ay_plan_forward) at the benchmark shape: the 2 GiB descriptor guard, 64
    # rotating deal-counter sets over ~76 launches and the arena's lifetime reuse all act at this size
    B = 64
    x64 = x1.to(dev).repeat(B, 1, 1, 1).contiguous()
    for rep in range(2):                                             # twice: the second pass reuses arena and counter sets
        out64 = m.forward_device(x64)
        assert out64.shape == (B, m.num_boxes(S), 5 + C_)
        assert bool(torch.isfinite(out64).all()), "non-finite rows"
        same = (out64 == out1[0:1]).all(dim=2).all(dim=1)
        assert bool(same.all()), f"pass {rep}: copies {torch.nonzero(~same).flatten().tolist()} differ from the batch-1 result"
    res = ay.non_max_suppression(out64.clone(), 0.5, 0.4)
    for b in range(1, B):
        assert np.array_equal(res.keep_idx[b], res.keep_idx[0]) and torch.equal(res[b], res[0])
    agree = parity.detection_agreement(z["nms_keep0"], z["nms_rows0"], res.keep_idx[0], res[0].cpu().numpy())
    s = parity.summarize([agree])
    print(f"configs[1] {precision} vs reference fixture:", s)
    assert s["keep_match"] >= KEEP_MATCH_MIN and s["extra_heads"] <= EXTRA_HEADS_MAX, s
    assert s["max_box_rel"] <= BOX_REL_MAX and s["max_dconf"] <= DCONF_MAX and agree["cls_equal"], s
    # decoded rows of the fixture's sample: confidences / classes within bf16 noise, boxes relative to their size
    got = out1[0].cpu().numpy()[z["out_rows"]]
    ref = z["out_sel"][0]
    dconf = np.abs(got[:, 4:] - ref[:, 4:])
    rel = np.abs(got[:, :4] - ref[:, :4]) / np.maximum(1.0, ref[:, 2:4].max(-1, keepdims=True))
    print("decoded sample: dconf q99 %.4f max %.4f, box rel q99.9 %.4f" % (np.quantile(dconf, 0.99), dconf.max(), np.quantile(rel, 0.999)))
    assert np.quantile(dconf, 0.99) <= DCONF_Q99 and np.quantile(rel, 0.999) <= REL_Q999


def test_hip_graph_of_a_detection_step_replays_after_eager_steps(tmp_cfg_dir):
    """One detection step (native plan forward + decode + merge-NMS) captured as a HIP graph (torch.cuda.CUDAGraph on a side
    stream), replayed UNFENCED -- a bare `g.replay()` followed by a device synchronisation, the form that failed in round 2 -- after
    interleaved eager steps on other inputs that use the same arena, output slot and NMS buffers: same bytes as the eager step on the
    same input.

    Record.  Rounds 1-2 saw GPU memory faults and inconsistent replays here.  The retained failing logs
    (gpurun_out/r2/graph{1,2,4,6}.log) all predate the two product fixes of that afternoon: a candidate counter that does not start
    at zero no longer turns into key writes past an image's array (`nms_filter_kernel`), and an item id fetched from a work counter
    is clamped to the launch's own range.  Three isolating experiments without torch or product kernels then cleared the runtime:
    every wait covers a replayed graph (scripts/micro/graph_sync.hip), a kernel behind a replay sees its writes (graph_coherence.hip),
    and a replay sees eager writes to its inputs while its nodes see each other's writes with every L2 warmed on older contents
    (graph_input_coherence.hip, round 4: 0 of 32 768 workgroup-results wrong).  Since round 4 a captured step holds kernel nodes of this
    library only (the candidate counters are zeroed by a kernel, not a memset node).  AY_TEST_GRAPH_WAIT=fence replays through
    `utils.graph_replay` (replay + `ay_stream_fence`); the other values select the bare waits of the round-2 comparison."""
    from amyloid_yolo_paper_amd.utils import nms_device
    dev = torch.device("cuda", 0)
    m, _ = build_models(3, tmp_cfg_dir, dev, "bf16")
    S, B = 256, 2
    xs = [torch.from_numpy(gc.model_inputs(S, B, start)).to(dev) for start in (0, 3, 5)]

    def step(x):
        out = m.forward_device(x, out_slot=0)
        return nms_device(out, 0.5, 0.4, 512, slot=7)

    def snapshot(res):
        return [t.clone() for t in res]

    ref = [snapshot(step(x)) for x in xs]          # eager results per input (also the warm-up: symbols, plans, buffers)
    torch.cuda.synchronize()
    static_x = xs[0].clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        res = step(static_x)
    torch.cuda.synchronize()
    for k in (1, 2, 0, 1):
        eager = snapshot(step(xs[2 - k if k != 1 else 1]))   # eager steps in between (they rotate the same counter sets and buffers)
        del eager
        static_x.copy_(xs[k])
        wait = os.environ.get("AY_TEST_GRAPH_WAIT", "devsync")
        if wait == "fence":
            ay.graph_replay(g)                  # replay + event fence, no host wait
        else:
            g.replay()
        if wait in ("devsync", "both"):
            torch.cuda.synchronize()
        if wait in ("tolist", "both"):
            res[2].tolist()                     # blocking device-to-host copy on the launch stream
        if wait == "streamsync":
            torch.cuda.current_stream().synchronize()
        if wait == "event":
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
        if wait == "waitevent":                 # no host wait at all: an event recorded behind the replay, waited for by the same stream
            ev = torch.cuda.Event()
            ev.record()
            torch.cuda.current_stream().wait_event(ev)
        rows, keep, count, cand = res
        rrows, rkeep, rcount, rcand = ref[k]
        if os.environ.get("AY_TEST_GRAPH_PRINT") == "1":
            print(f"replay on input {k}: count {count.tolist()} cand {cand.tolist()}; eager {rcount.tolist()} {rcand.tolist()}; all refs {[r[2].tolist() for r in ref]}")
        assert torch.equal(count, rcount) and torch.equal(cand, rcand), f"replay on input {k}: counts differ from the eager step"
        for b in range(B):
            n = int(rcount[b])     # rows beyond an image's own count are leftovers of earlier calls
            assert torch.equal(rows[b, :n], rrows[b, :n]) and torch.equal(keep[b, :n], rkeep[b, :n]), f"replay on input {k}, image {b}"


def test_configs2_train_b32_100_steps():
    """BASELINE.json configs[2]: random-init YOLOv3 (3 classes), batch 32, synthetic boxes, 100 optimiser steps on the bf16 MFMA
    path with the step bench.py --mode train times (flat gradient buffer, ay_adam_flat): every loss finite, the loss falls
    (mean of the last 10 steps below a third of the first), filters are re-packed once per step, and the step's memory stays
    bounded -- at 1024^2 (100 steps too) below 60 GB (round 1 peaked at ~219 GB through per-step allocations)."""
    from amyloid_yolo_paper_amd import cfg_gen, synth
    from amyloid_yolo_paper_amd.models import Darknet
    from amyloid_yolo_paper_amd.parallel import FlatAdam, FlatGradReducer
    from amyloid_yolo_paper_amd.utils import weights_init_normal
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = Darknet(cfg_gen.write_cfg(3), precision="bf16").to(dev)
    model.apply(weights_init_normal)
    model.train()
    model.collect_metrics = False
    red = FlatGradReducer(model.parameters(), n_buckets=4).attach(model)
    opt = FlatAdam(red)
    for S, steps in ((416, 100), (1024, 100)):   # configs[2] as written: 100 optimiser steps at B=32, at the reference's 416 AND at 1024^2
        B = 32
        x = torch.from_numpy(synth.synth_tiles(8, S, start=100)).to(dev).repeat(4, 1, 1, 1).contiguous()
        tg = torch.from_numpy(synth.synth_targets(B, 3, seed=77, grid=S // 8)).to(dev)
        torch.cuda.reset_peak_memory_stats()
        losses = []
        for _ in range(steps):
            red.begin()
            loss, _ = model.train_step_device(x, tg)
            loss.backward()
            red.all_reduce(average=False)
            opt.step()
            red.zero()
            losses.append(loss)
        lv = np.array([float(v.item()) for v in losses])
        assert np.isfinite(lv).all(), lv
        if S == 416:      # from the random init
            assert lv[-10:].mean() < lv[0] / 3.0, (lv[0], lv[-10:].mean())
        else:             # the same model and optimiser state carried on at the larger size: keeps falling
            assert lv[-10:].mean() < lv[:10].mean(), (lv[:10].mean(), lv[-10:].mean())
        peak = torch.cuda.max_memory_allocated() / 1e9
        print(f"configs[2] S={S}: loss {lv[0]:.1f} -> {lv[-1]:.1f}, peak HBM {peak:.1f} GB")
        if S == 1024:
            assert peak < 60.0, peak
        model._train_ctx.clear()
        torch.cuda.empty_cache()
