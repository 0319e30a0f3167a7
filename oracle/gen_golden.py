"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference, CPU).

TEST INFRASTRUCTURE.  Run in the build container only (the reference never travels):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Inputs come from tests/golden_cases.py (seeded); only the reference's OUTPUTS are stored.
Nothing from the reference tree is copied: it is imported, called and its results saved.
"""
import hashlib
import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AY_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(1, REF)

import golden_cases as gc  # noqa: E402
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth  # noqa: E402

import models as ref_models  # noqa: E402  (reference)
from utils import utils as ref_utils  # noqa: E402  (reference)

OUT = os.path.join(REPO, "tests", "golden")
torch.manual_seed(0)
torch.set_num_threads(8)


def save(name, **arrays):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def keep_indices(pred_corners_b, rows):
    """original row of every output row: match the untouched (conf, cls_conf, cls_pred) triple."""
    conf = pred_corners_b[:, 4]
    cls_conf = pred_corners_b[:, 5:].max(1)
    cls_pred = pred_corners_b[:, 5:].argmax(1).astype(np.float32)
    idx = []
    for r in rows:
        m = np.nonzero((conf == r[4]) & (cls_conf == r[5]) & (cls_pred == r[6]))[0]
        assert m.size == 1, "synthetic rows must be unique"
        idx.append(m[0])
    return np.asarray(idx, np.int64)


def gen_kats():
    t = torch.tensor
    iou = ref_utils.bbox_iou(t([[100., 100, 200, 200]]), t([[150., 150, 200, 200], [201, 201, 300, 300], [100, 100, 200, 200]]))
    iou_c = ref_utils.bbox_iou(t([[5., 5, 4, 4]]), t([[6., 6, 4, 4]]), x1y1x2y2=False)
    whiou = ref_utils.bbox_wh_iou(t([3., 4.]), t([[3., 4], [6, 2], [1, 1]]))
    b1, b2 = gc.iou_inputs()
    iou_rand = ref_utils.bbox_iou(t(b1), t(b2))
    iou_rand_c = ref_utils.bbox_iou(t(b1), t(b2), x1y1x2y2=False)
    iou_bcast = ref_utils.bbox_iou(t(b1[:1]), t(b2))
    xyxy = ref_utils.xywh2xyxy(t(b1))
    # decode KAT
    layer = ref_models.YOLOLayer([(10, 13), (16, 30), (33, 23)], 2, 64)
    p = torch.zeros(1, 21, 2, 2)
    p[0, 0, 1, 0] = 1.0
    p[0, 7 + 2, 0, 1] = 0.5
    p[0, 14 + 4, 1, 1] = 2.0
    dec, _ = layer(p, None, 64)
    # NMS KAT
    rows = np.array([[100, 100, 50, 50, .9, .2, .8], [104, 102, 50, 50, .6, .1, .9], [102, 101, 52, 48, .95, .7, .3],
                     [300, 300, 40, 40, .7, .6, .1], [100, 100, 50, 50, .4, .1, .9], [108, 100, 50, 50, .55, .3, .95]], np.float32)
    nms = ref_utils.non_max_suppression(torch.from_numpy(rows[None].copy()), 0.5, 0.4)[0]
    rb = ref_utils.rescale_boxes(torch.from_numpy(np.array([[10., 20, 200, 300, .9, .8, 1], [50, 60, 70, 80, .5, .5, 0]], np.float32)), 416, (1536, 1024))
    save("kat", iou=iou.numpy(), iou_c=iou_c.numpy(), whiou=whiou.numpy(), iou_rand=iou_rand.numpy(),
         iou_rand_c=iou_rand_c.numpy(), iou_bcast=iou_bcast.numpy(), xyxy=xyxy.numpy(), decode=dec.numpy(),
         nms_in=rows, nms_out=nms.numpy(), rescale=rb.numpy())


def gen_nms():
    for name, _rows, _cands, _C, _seed in gc.NMS_CASES:
        pred, conf_t, nms_t = gc.nms_case_inputs(name)
        for b in range(pred.shape[0]):  # the reference's argsort leaves tie order unspecified (utils/utils.py:255)
            c = pred[b][pred[b, :, 4] >= conf_t]
            sc = c[:, 4] * c[:, 5:].max(1)
            assert np.unique(sc).size == sc.size or name == "dups_crossclass", f"score tie in {name}: pick another seed"
        tp = torch.from_numpy(pred.copy())
        out = ref_utils.non_max_suppression(tp, conf_t, nms_t)
        corners = tp.numpy()  # mutated in place to corners by the reference
        arrays = {"conf_thres": np.float32(conf_t), "nms_thres": np.float32(nms_t), "corners0": corners[0, :8].copy()}
        for b, o in enumerate(out):
            if o is None:
                arrays[f"n{b}"] = np.int64(0)
                continue
            o = o.numpy()
            arrays[f"n{b}"] = np.int64(o.shape[0])
            arrays[f"rows{b}"] = o
            arrays[f"keep{b}"] = keep_indices(corners[b], o)
        save("nms_" + name, **arrays)


def build_ref_model(C, S, tmpdir):
    cfg = cfg_gen.write_cfg(C, tmpdir)
    defs = parse_config.parse_model_config(cfg)
    params = synth.synth_params(defs, seed=7)
    wpath = os.path.join(tmpdir, f"synth_c{C}.weights")
    if not os.path.exists(wpath):
        synth.write_darknet_weights(wpath, defs, params, seen=12345)
    model = ref_models.Darknet(cfg, img_size=S)
    model.load_darknet_weights(wpath)
    return model, cfg, wpath


def gen_models(tmpdir):
    for name, C, S, B, start in gc.MODEL_CASES:
        model, cfg, wpath = build_ref_model(C, S, tmpdir)
        model.eval()
        x = torch.from_numpy(gc.model_inputs(S, B, start))
        feats = {}
        hooks = []
        for i, m in enumerate(model.module_list):
            if model.module_defs[i]["type"] == "convolutional":
                hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: feats.__setitem__(i, out.detach())))
        with torch.no_grad():
            out = model(x)
        for h in hooks:
            h.remove()
        arrays = {}
        N = out.shape[1]
        if N <= 12000:
            arrays["out"] = out.numpy()
        else:
            sel = np.random.Generator(np.random.PCG64(99)).choice(N, 2048, replace=False)
            sel.sort()
            arrays["out_rows"] = sel.astype(np.int64)
            arrays["out_sel"] = out.numpy()[:, sel]
        # per conv layer: mean, abs-max and 64 sampled values (flattened NCHW index)
        stat_idx, stat_mean, stat_amax, samp_idx, samp_val = [], [], [], [], []
        for i in sorted(feats):
            f = feats[i].numpy().reshape(-1)
            ridx = np.random.Generator(np.random.PCG64(500 + i)).integers(0, f.size, 64)
            stat_idx.append(i); stat_mean.append(f.mean(dtype=np.float64)); stat_amax.append(np.abs(f).max())
            samp_idx.append(ridx); samp_val.append(f[ridx])
        arrays.update(layer_idx=np.asarray(stat_idx), layer_mean=np.asarray(stat_mean, np.float32),
                      layer_amax=np.asarray(stat_amax, np.float32), samp_idx=np.stack(samp_idx), samp_val=np.stack(samp_val))
        if S <= 160:  # full head tensors (pre-decode) for the small cases
            for j, li in enumerate([i for i, d in enumerate(model.module_defs) if d["type"] == "yolo"]):
                arrays[f"head{j}"] = feats[li - 1].numpy()
        nms = ref_utils.non_max_suppression(out.clone(), 0.5, 0.4)
        corners = out.clone()
        corners[..., :4] = ref_utils.xywh2xyxy(corners[..., :4])
        for b, o in enumerate(nms):
            arrays[f"nms_n{b}"] = np.int64(0 if o is None else o.shape[0])
            if o is not None:
                arrays[f"nms_rows{b}"] = o.numpy()
                arrays[f"nms_keep{b}"] = keep_indices(corners[b].numpy(), o.numpy())
        save("model_" + name, **arrays)
        if name == "c2_s64_b2":
            # .weights round trip through the reference writer must be byte-identical to our writer
            rt = os.path.join(tmpdir, "roundtrip.weights")
            model.save_darknet_weights(rt)
            h1 = hashlib.sha256(open(wpath, "rb").read()).hexdigest()
            h2 = hashlib.sha256(open(rt, "rb").read()).hexdigest()
            assert h1 == h2, "reference round trip differs from our writer"
            sd = model.state_dict()
            save("weights_c2", sha256=np.frombuffer(bytes.fromhex(h1), np.uint8), nbytes=np.int64(os.path.getsize(wpath)),
                 n_state_keys=np.int64(len(sd)), seen=np.int64(model.seen),
                 state_keys=np.array(list(sd.keys())), state_numel=np.array([v.numel() for v in sd.values()], np.int64))


def gen_train(tmpdir):
    for name, C, S, B, seed in gc.TRAIN_CASES:
        model, cfg, wpath = build_ref_model(C, S, tmpdir)
        model.train()
        x = torch.from_numpy(gc.model_inputs(S, B, 10))
        tg = torch.from_numpy(gc.train_targets(B, C, S, seed))
        loss, out = model(x, tg)
        loss.backward()
        arrays = {"loss": np.float32(loss.item()), "targets": tg.numpy(), "out": out.numpy()}
        keys = ["loss", "x", "y", "w", "h", "conf", "cls", "cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj", "grid_size"]
        arrays["metrics"] = np.array([[yl.metrics[k] for k in keys] for yl in model.yolo_layers], np.float64)
        arrays["metric_keys"] = np.array(keys)
        for li in (0, 1, 2, 42, 73, 80, 81, 93, 104, 105):
            conv = model.module_list[li][0]
            g = conv.weight.grad.numpy()
            arrays[f"gw{li}"] = g if g.size <= 300000 else g.reshape(-1)[:: max(1, g.size // 65536)].copy()
            if conv.bias is not None:
                arrays[f"gb{li}"] = conv.bias.grad.numpy()
            else:
                bn = model.module_list[li][1]
                arrays[f"ggamma{li}"] = bn.weight.grad.numpy()
                arrays[f"gbeta{li}"] = bn.bias.grad.numpy()
                arrays[f"rmean{li}"] = bn.running_mean.numpy().copy()
                arrays[f"rvar{li}"] = bn.running_var.numpy().copy()
        save(name, **arrays)
        # build_targets alone, on the stride-8 head of this case
        yl = model.yolo_layers[-1]
        G = S // 8
        rng = np.random.Generator(np.random.PCG64(seed + 100))
        pb = rng.uniform(0, G, (B, 3, G, G, 4)).astype(np.float32)
        pc = rng.uniform(0, 1, (B, 3, G, G, C)).astype(np.float32)
        sa = torch.tensor([(a[0] / 8.0, a[1] / 8.0) for a in yl.anchors])
        bt = ref_utils.build_targets(torch.from_numpy(pb), torch.from_numpy(pc), tg, sa, 0.5)
        names = ["iou_scores", "class_mask", "obj_mask", "noobj_mask", "tx", "ty", "tw", "th", "tcls", "tconf"]
        save("bt_" + name, anchors=sa.numpy(), **{n: v.numpy() for n, v in zip(names, bt)})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["kat", "nms", "models", "train"]
    tmp = os.environ.get("AY_TMP", os.path.join(tempfile.gettempdir(), "ay_golden"))
    os.makedirs(tmp, exist_ok=True)
    if "kat" in which:
        gen_kats()
    if "nms" in which:
        gen_nms()
    if "models" in which:
        gen_models(tmp)
    if "train" in which:
        gen_train(tmp)
