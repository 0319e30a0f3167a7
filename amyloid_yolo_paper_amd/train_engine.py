"""Training step of the Darknet graph on the HIP path: ``model(x, targets)`` -> ``(loss, outputs)`` with
``loss.backward()`` filling ``.grad`` of every parameter, as in the reference (``models.py:237-255`` forward,
autograd backward, ``train.py:113-119``).

There is no autograd tape over torch ops here: the forward runs the layer plan through the C ABI and keeps the
activations; the backward walks the plan in reverse calling the dgrad / wgrad / BN-backward / loss kernels.  A single
``torch.autograd.Function`` node ties that to PyTorch's optimiser plumbing (``loss.backward()``, ``optimizer.step()``).

fp32 reference-precision path (NCHW): this is the path the golden training fixtures pin (loss, per-layer metrics,
gradients, BN running statistics).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, check, ptr

METRIC_KEYS = ["loss", "x", "y", "w", "h", "conf", "cls", "cls_acc", "recall50", "recall75", "precision", "conf_obj",
               "conf_noobj", "grid_size"]


class _State:
    """activations kept between forward and backward"""

    def __init__(self):
        self.val = {}      # layer -> output tensor (or ("up", src) lazy upsample)
        self.conv = {}     # layer -> dict(x=input tensor, z=raw conv out, y=block out, mean, invstd, desc, src)
        self.dhead = {}    # yolo layer -> gradient w.r.t. its head tensor
        self.route = {}    # route layer -> list of (src layer, channels, up)


def _ws(model, nbytes, dev):
    ws = getattr(model, "_loss_ws", None)
    if ws is None or ws.numel() < nbytes or ws.device != dev:
        ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
        model._loss_ws = ws
    return ws


def train_forward(model, x, targets):
    """Runs the plan in fp32 NCHW.  Returns (loss tensor [device scalar] | None, outputs [B,N,5+C] device, state)."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
    B, _, S, S2 = x.shape
    assert S == S2 and S % 32 == 0
    training = model.training
    graph = model._graph
    stt = _State()
    val = stt.val
    Ccls = model.yolo_layers[0].num_classes
    N = model.num_boxes(S)
    out = torch.empty(B, N, 5 + Ccls, device=dev, dtype=torch.float32)
    tg = None if targets is None else targets.detach().to(device=dev, dtype=torch.float32).contiguous()

    def size_of(i):
        return S >> graph[i]["log2_down"] if i >= 0 else S

    def resolve(i):
        v = val[i]
        if isinstance(v, tuple):  # lazy nearest x2 upsample
            src = resolve(v[1])
            c, h = graph[i]["channels"], size_of(i)
            o = torch.empty(B, c, h, h, device=dev, dtype=torch.float32)
            check(L.ay_copy_channels_f32(ptr(src), ptr(o), B, c, c, 0, h, h, 1, st), "ay_copy_channels_f32")
            val[i] = o
            return o
        return v

    row = 0
    loss = None
    sums_all = []
    for i, e in enumerate(graph):
        t = e["type"]
        if t == "convolutional":
            m = model.module_list[i]
            conv = m[0]
            src = x if e["src"] < 0 else resolve(e["src"])
            hin, hout = size_of(e["src"]), size_of(i)
            cout = e["cout"]
            w = conv.weight.detach()
            d = ConvDesc(B, e["cin"], cout, hin, hin, hout, hout, e["k"], e["stride"], 0, 0, cout)
            z = torch.empty(B, cout, hout, hout, device=dev, dtype=torch.float32)
            ones, zeros = model._unit(cout, dev)
            rec = dict(x=src, desc=d, src=e["src"], w=w)
            if e["bn"]:
                bn = m[1]
                if training:
                    check(L.ay_conv_fwd_f32_valu(C.byref(d), ptr(src), e["cin"], 0, None, ptr(w), ptr(ones), ptr(zeros), None, ptr(z), st),
                          "ay_conv_fwd_f32")
                    y = torch.empty_like(z)
                    mean = torch.empty(cout, device=dev, dtype=torch.float32)
                    invstd = torch.empty(cout, device=dev, dtype=torch.float32)
                    check(L.ay_bn_train_fwd_f32(ptr(z), ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean),
                                                ptr(bn.running_var), C.c_float(bn.momentum), C.c_float(bn.eps), int(e["leaky"]), ptr(y),
                                                ptr(mean), ptr(invstd), B, cout, hout * hout, st), "ay_bn_train_fwd_f32")
                    bn.num_batches_tracked += 1
                    rec.update(z=z, y=y, mean=mean, invstd=invstd)
                else:  # eval-mode BN with a loss: affine from the running statistics (no backward through this mode)
                    scale = torch.empty(cout, device=dev, dtype=torch.float32)
                    shift = torch.empty(cout, device=dev, dtype=torch.float32)
                    check(L.ay_fold_bn(ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean), ptr(bn.running_var), None,
                                       C.c_float(bn.eps), ptr(scale), ptr(shift), cout, cout, st), "ay_fold_bn")
                    d.leaky = int(e["leaky"])
                    check(L.ay_conv_fwd_f32_valu(C.byref(d), ptr(src), e["cin"], 0, None, ptr(w), ptr(scale), ptr(shift), None, ptr(z), st),
                          "ay_conv_fwd_f32")
                    d.leaky = 0
                    y = z
                    rec.update(z=None, y=y, keep=(scale, shift))
            else:
                bias = conv.bias.detach()
                check(L.ay_conv_fwd_f32_valu(C.byref(d), ptr(src), e["cin"], 0, None, ptr(w), ptr(ones), ptr(bias), None, ptr(z), st),
                      "ay_conv_fwd_f32")
                y = z
                rec.update(z=z, y=y)
            stt.conv[i] = rec
            val[i] = y
        elif t == "shortcut":
            a, b_ = resolve(e["a"]), resolve(e["b"])
            o = torch.empty_like(a)
            check(L.ay_add_f32(ptr(a), ptr(b_), ptr(o), a.numel(), st), "ay_add_f32")
            val[i] = o
        elif t == "upsample":
            val[i] = ("up", e["src"])
        elif t == "route":
            srcs = e["srcs"]
            if len(srcs) == 1 and not isinstance(val[srcs[0]], tuple):
                val[i] = val[srcs[0]]
                stt.route[i] = [(srcs[0], graph[srcs[0]]["channels"], 0)]
            else:
                h, ctot = size_of(i), e["channels"]
                o = torch.empty(B, ctot, h, h, device=dev, dtype=torch.float32)
                c0, parts = 0, []
                for j in srcs:
                    up = isinstance(val[j], tuple)
                    base = val[j][1] if up else j
                    s = resolve(base)
                    cj = graph[j]["channels"]
                    check(L.ay_copy_channels_f32(ptr(s), ptr(o), B, cj, ctot, c0, h, h, int(up), st), "ay_copy_channels_f32")
                    parts.append((base, cj, int(up)))
                    c0 += cj
                val[i] = o
                stt.route[i] = parts
        elif t == "yolo":
            y = model.module_list[i][0]
            head = resolve(e["src"])
            G = size_of(i)
            anchors = (C.c_float * (2 * y.num_anchors))(*[float(v) for a in y.anchors for v in a])
            check(L.ay_yolo_decode(ptr(head), 0, ptr(out), B, y.num_anchors, y.num_classes, G, S, anchors, N, row, st), "ay_yolo_decode")
            y.grid_size, y.img_dim = G, S
            row += y.num_anchors * G * G
            if tg is not None:
                dhead = torch.empty_like(head)
                sums = torch.empty(16, device=dev, dtype=torch.float32)
                nb = L.ay_yolo_loss_workspace_bytes(B, y.num_anchors, y.num_classes, G)
                ws = _ws(model, nb, dev)
                check((L.ay_yolo_loss_giou_fwd_bwd if getattr(model, 'box_loss', 'mse') == 'giou' else L.ay_yolo_loss_fwd_bwd)(ptr(head), ptr(tg), tg.shape[0], B, y.num_anchors, y.num_classes, G, S, anchors,
                                             C.c_float(y.ignore_thres), C.c_float(1.0), ptr(dhead), ptr(sums), ptr(ws), ws.numel(), st),
                      "ay_yolo_loss_fwd_bwd")
                stt.dhead[i] = dhead
                sums_all.append((y, sums, G))
            val[i] = head
    if tg is not None:
        # six terms per layer from the device sums (scalar glue on the device; the sums came from the HIP loss kernel)
        allsums = torch.stack([s for _, s, _ in sums_all])  # [3,16]
        n_obj, n_noobj = allsums[:, 7], allsums[:, 8]
        lx, ly, lw, lh = (allsums[:, k] / n_obj for k in range(4))
        lconf = allsums[:, 4] / n_obj + 100.0 * allsums[:, 5] / n_noobj
        lcls = allsums[:, 6] / (n_obj * Ccls)
        per_layer = lx + ly + lw + lh + lconf + lcls
        loss = per_layer.sum()
        h = torch.stack([per_layer, lx, ly, lw, lh, lconf, lcls, 100.0 * allsums[:, 9] / n_obj, allsums[:, 13] / (n_obj + 1e-16),
                         allsums[:, 14] / (n_obj + 1e-16), allsums[:, 13] / (allsums[:, 12] + 1e-16), allsums[:, 10] / n_obj,
                         allsums[:, 11] / n_noobj], 1).cpu().numpy()  # the one host sync of the step (reference: 39)
        for li, (y, _, G) in enumerate(sums_all):
            y.metrics = {k: float(h[li, j]) for j, k in enumerate(METRIC_KEYS[:-1])}
            y.metrics["grid_size"] = G
    return loss, out, stt


def train_backward(model, stt, grad_scale=1.0):
    """Reverse walk of the plan; returns gradients in ``model.parameters()`` order."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    graph = model._graph
    dval = {}

    def acc(j, t):
        if j < 0:
            return
        if j not in dval:
            dval[j] = t.clone()
        else:
            check(L.ay_accumulate_f32(ptr(dval[j]), ptr(t), t.numel(), st), "ay_accumulate_f32")

    grads = {}
    for i in range(len(graph) - 1, -1, -1):
        e = graph[i]
        t = e["type"]
        if t == "yolo":
            if i in stt.dhead:
                acc(e["src"], stt.dhead[i])
        elif t == "route":
            if i not in dval:
                continue
            d = dval.pop(i)
            parts = stt.route[i]
            B, ctot, h, _ = d.shape
            if len(parts) == 1 and parts[0][2] == 0 and parts[0][1] == ctot:
                acc(parts[0][0], d)
            else:
                c0 = 0
                for base, cj, up in parts:
                    hs = h >> up
                    if base not in dval:
                        dval[base] = torch.zeros(B, cj, hs, hs, device=d.device, dtype=torch.float32)
                    check(L.ay_slice_accumulate_f32(ptr(d), ptr(dval[base]), B, cj, ctot, c0, h, h, up, st), "ay_slice_accumulate_f32")
                    c0 += cj
        elif t == "upsample":
            if i in dval:  # an upsample consumed by something other than a route
                d = dval.pop(i)
                B, c, h, _ = d.shape
                if e["src"] not in dval:
                    dval[e["src"]] = torch.zeros(B, c, h // 2, h // 2, device=d.device, dtype=torch.float32)
                check(L.ay_slice_accumulate_f32(ptr(d), ptr(dval[e["src"]]), B, c, c, 0, h, h, 1, st), "ay_slice_accumulate_f32")
        elif t == "shortcut":
            if i not in dval:
                continue
            d = dval.pop(i)
            acc(e["a"], d)
            acc(e["b"], d)
        elif t == "convolutional":
            if i not in dval:
                continue
            dy = dval.pop(i)
            if getattr(model, "_keep_dval", None) is not None:
                model._keep_dval[i] = dy.clone()  # debugging hook: gradient w.r.t. this block's output
            rec = stt.conv[i]
            m = model.module_list[i]
            conv = m[0]
            d = rec["desc"]
            if e["bn"]:
                if rec["z"] is None:
                    raise RuntimeError("backward through eval-mode BatchNorm is not supported: call model.train()")
                bn = m[1]
                dz = torch.empty_like(dy)
                dg = torch.empty(e["cout"], device=dy.device, dtype=torch.float32)
                db = torch.empty(e["cout"], device=dy.device, dtype=torch.float32)
                check(L.ay_bn_train_bwd_f32(ptr(dy), ptr(rec["y"]), ptr(rec["z"]), ptr(bn.weight.detach()), ptr(rec["mean"]),
                                            ptr(rec["invstd"]), int(e["leaky"]), ptr(dz), ptr(dg), ptr(db), d.batch, e["cout"],
                                            d.hout * d.wout, st), "ay_bn_train_bwd_f32")
                grads[bn.weight] = dg
                grads[bn.bias] = db
            else:
                dz = dy
                db = torch.empty(e["cout"], device=dy.device, dtype=torch.float32)
                check(L.ay_bias_grad_f32(ptr(dz), ptr(db), d.batch, e["cout"], d.hout * d.wout, st), "ay_bias_grad_f32")
                grads[conv.bias] = db
            dw = torch.empty_like(rec["w"])
            check(L.ay_conv_wgrad_f32(C.byref(d), ptr(rec["x"]), ptr(dz), ptr(dw), st), "ay_conv_wgrad_f32")
            grads[conv.weight] = dw
            j = rec["src"]
            if j >= 0:
                first = j not in dval
                if first:
                    dval[j] = torch.empty_like(rec["x"])
                check(L.ay_conv_dgrad_f32(C.byref(d), ptr(dz), ptr(rec["w"]), ptr(dval[j]), 0 if first else 1, st), "ay_conv_dgrad_f32")
    out = []
    for p in model.parameters():
        g = grads.get(p)
        if g is None:
            g = torch.zeros_like(p)
        elif grad_scale != 1.0:
            g = g * grad_scale
        out.append(g)
    return out


class TrainStep(torch.autograd.Function):
    """(loss, outputs) = TrainStep.apply(model, x, targets, *model.parameters())"""

    @staticmethod
    def forward(ctx, model, x, targets, *params):
        loss, out, stt = train_forward(model, x, targets)
        ctx.model, ctx.stt = model, stt
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        gs = float(grad_loss.item()) if grad_loss is not None else 1.0
        grads = train_backward(ctx.model, ctx.stt, gs)
        ctx.stt = None
        return (None, None, None) + tuple(grads)
