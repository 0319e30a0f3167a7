"""Training step on the bf16 MFMA path (``Darknet(precision="bf16")``): same contract as ``train_engine`` (fp32), but

* convolutions, their data gradients (the forward kernel on re-packed filters) and their weight gradients
  (``ay_conv_wgrad_bf16``) run on the matrix cores over blocked-bf16 tensors;
* train-mode BatchNorm is a statistics pass + an apply pass around the raw convolution output, the shortcut add fused
  into the apply pass; its backward recomputes the pre-activation from the saved raw output;
* the stem runs through the same kernels (the fp32 image becomes one zero-padded 16-channel bf16 plane); the YOLO loss
  stays on the fp32 kernel;
* parameter gradients are fp32 (``.grad`` of the fp32 master parameters), activations / activation gradients bf16.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr
from .train_engine import METRIC_KEYS, _ws


def _pad(v, m):
    return (v + m - 1) // m * m


class _State:
    def __init__(self):
        self.val = {}
        self.conv = {}
        self.dhead = {}
        self.route = {}
        self.B = self.S = 0


def train_forward_bf16(model, x, targets):
    L = _lib.lib()
    st = _lib.stream_ptr()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
    B, _, S, S2 = x.shape
    assert S == S2 and S % 32 == 0
    assert model.training, "the bf16 training path implements train-mode BatchNorm only"
    graph = model._graph
    stt = _State()
    stt.B, stt.S = B, S
    val = stt.val
    Ccls = model.yolo_layers[0].num_classes
    N = model.num_boxes(S)
    out = torch.empty(B, N, 5 + Ccls, device=dev, dtype=torch.float32)
    tg = None if targets is None else targets.detach().to(device=dev, dtype=torch.float32).contiguous()

    def size_of(i):
        return S >> graph[i]["log2_down"] if i >= 0 else S

    def blocked(c, h, dtype=torch.bfloat16, pad=16):
        return torch.empty(B, _pad(c, pad) // 16, h, h, 16, device=dev, dtype=dtype)

    def resolve(i):
        v = val[i]
        if isinstance(v, tuple):
            src = resolve(v[1])
            c, h = graph[i]["channels"], size_of(i)
            o = blocked(c, h)
            check(L.ay_concat_upsample_bf16(ptr(src), c, 1, None, 0, ptr(o), B, h, h, st), "ay_concat_upsample_bf16")
            val[i] = o
            return o
        return v

    row = 0
    sums_all = []
    n_layers = len(graph)
    skip_next = False
    for i, e in enumerate(graph):
        t = e["type"]
        if t == "convolutional":
            m = model.module_list[i]
            conv = m[0]
            hin, hout = size_of(e["src"]), size_of(i)
            cout, cin, k = e["cout"], e["cin"], e["k"]
            w = conv.weight.detach()
            if e["src"] < 0:
                # ---- stem: the fp32 image becomes ONE zero-padded 16-channel bf16 plane, the filters get zero input
                # channels 3..15, and the layer runs through the same MFMA kernels as every other layer
                assert e["bn"] and cin <= 16
                xb = blocked(16, hin)
                check(L.ay_nchw_f32_to_blocked_bf16(ptr(x), ptr(xb), B, cin, hin, hin, st), "ay_nchw_f32_to_blocked_bf16")
                w16 = torch.zeros(cout, 16, k, k, device=dev, dtype=torch.float32)
                w16[:, :cin] = w
                src, stem_w, cin_eff = xb, w16, 16
            else:
                src, stem_w, cin_eff = resolve(e["src"]), None, cin
            cpad = _pad(cout, 32)
            is_head = not e["bn"]
            packed = torch.empty(L.ay_packed_weight_bytes(cpad, cin_eff, k), device=dev, dtype=torch.uint8)
            check(L.ay_pack_conv_weights_bf16(ptr(w if stem_w is None else stem_w), ptr(packed), cout, cpad, cin_eff, k, st), "ay_pack_conv_weights_bf16")
            ones, zeros = model._unit(cpad, dev)
            d = ConvDesc(B, cin_eff, cout, hin, hin, hout, hout, k, e["stride"], 0, int(is_head), cpad)
            rec = dict(kind="head" if is_head else "bn", x=src, desc=d, w=w, src=e["src"], cpad=cpad, stem=stem_w is not None)
            if is_head:
                shift = torch.zeros(cpad, device=dev, dtype=torch.float32)
                shift[:cout] = conv.bias.detach()
                zb = blocked(cout, hout, torch.float32, 32)
                check(L.ay_conv_fwd_bf16(C.byref(d), ptr(src), ptr(packed), ptr(ones), ptr(shift), None, ptr(zb), st), "ay_conv_fwd_bf16")
                head = torch.empty(B, cout, hout, hout, device=dev, dtype=torch.float32)
                check(L.ay_blocked_f32_to_nchw_f32(ptr(zb), ptr(head), B, cout, hout, hout, st), "ay_blocked_f32_to_nchw_f32")
                rec["keep"] = (shift, packed)
                stt.conv[i] = rec
                val[i] = head
                continue
            bn = m[1]
            z = blocked(cout, hout, pad=32)
            check(L.ay_conv_fwd_bf16(C.byref(d), ptr(src), ptr(packed), ptr(ones), ptr(zeros), None, ptr(z), st), "ay_conv_fwd_bf16")
            fuse = e["fuse_into_shortcut"]
            skip = resolve(graph[i + 1]["b"]) if fuse else None
            y = blocked(cout, hout, pad=32)
            mean = torch.empty(cout, device=dev, dtype=torch.float32)
            invstd = torch.empty(cout, device=dev, dtype=torch.float32)
            ws = torch.empty(2 * cout, device=dev, dtype=torch.float64)
            check(L.ay_bn_train_fwd_bf16(ptr(z), ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean), ptr(bn.running_var),
                                         C.c_float(bn.momentum), C.c_float(bn.eps), int(e["leaky"]), ptr(skip), ptr(y), ptr(mean), ptr(invstd),
                                         ptr(ws), B, cout, hout, hout, st), "ay_bn_train_fwd_bf16")
            bn.num_batches_tracked += 1
            rec.update(z=z, mean=mean, invstd=invstd, fused=fuse, keep=(packed, ws))
            stt.conv[i] = rec
            if fuse:
                val[i] = None          # never materialised: only the following shortcut uses it
                val[i + 1] = y
            else:
                val[i] = y
        elif t == "shortcut":
            if val.get(i) is None:
                raise NotImplementedError(f"layer {i}: shortcut whose first operand is not the preceding convolution")
        elif t == "upsample":
            val[i] = ("up", e["src"])
        elif t == "route":
            srcs = e["srcs"]
            if len(srcs) == 1 and not isinstance(val[srcs[0]], tuple):
                val[i] = val[srcs[0]]
                stt.route[i] = [(srcs[0], graph[srcs[0]]["channels"], 0)]
            else:
                assert len(srcs) == 2, "route with more than two sources"
                h = size_of(i)
                a, b_ = srcs
                up = isinstance(val[a], tuple)
                base = val[a][1] if up else a
                s1, s2 = resolve(base), resolve(b_)
                o = blocked(e["channels"], h)
                check(L.ay_concat_upsample_bf16(ptr(s1), graph[a]["channels"], int(up), ptr(s2), graph[b_]["channels"], ptr(o), B, h, h, st),
                      "ay_concat_upsample_bf16")
                val[i] = o
                stt.route[i] = [(base, graph[a]["channels"], int(up)), (b_, graph[b_]["channels"], 0)]
        elif t == "yolo":
            y = model.module_list[i][0]
            head = val[e["src"]]
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
    loss = None
    if tg is not None:
        allsums = torch.stack([s for _, s, _ in sums_all])
        n_obj, n_noobj = allsums[:, 7], allsums[:, 8]
        lx, ly, lw, lh = (allsums[:, k] / n_obj for k in range(4))
        lconf = allsums[:, 4] / n_obj + 100.0 * allsums[:, 5] / n_noobj
        lcls = allsums[:, 6] / (n_obj * Ccls)
        per_layer = lx + ly + lw + lh + lconf + lcls
        loss = per_layer.sum()
        h = torch.stack([per_layer, lx, ly, lw, lh, lconf, lcls, 100.0 * allsums[:, 9] / n_obj, allsums[:, 13] / (n_obj + 1e-16),
                         allsums[:, 14] / (n_obj + 1e-16), allsums[:, 13] / (allsums[:, 12] + 1e-16), allsums[:, 10] / n_obj,
                         allsums[:, 11] / n_noobj], 1).cpu().numpy()
        for li, (y, _, G) in enumerate(sums_all):
            y.metrics = {k: float(h[li, j]) for j, k in enumerate(METRIC_KEYS[:-1])}
            y.metrics["grid_size"] = G
    return loss, out, stt


def train_backward_bf16(model, stt, grad_scale=1.0):
    L = _lib.lib()
    st = _lib.stream_ptr()
    graph = model._graph
    B, S = stt.B, stt.S
    dev = torch.device("cuda", torch.cuda.current_device())
    dval = {}

    def size_of(i):
        return S >> graph[i]["log2_down"] if i >= 0 else S

    def acc(j, t):
        """accumulate the blocked-bf16 gradient t into layer j's output gradient"""
        if j < 0:
            return
        if j not in dval:
            dval[j] = t.clone()
        else:
            check(L.ay_accumulate_bf16(ptr(dval[j]), ptr(t), t.numel(), st), "ay_accumulate_bf16")

    grads = {}
    for i in range(len(graph) - 1, -1, -1):
        e = graph[i]
        t = e["type"]
        if t == "yolo":
            continue  # dhead is consumed by the head convolution below
        if t == "route":
            if i not in dval:
                continue
            d = dval.pop(i)
            parts = stt.route[i]
            h = size_of(i)
            ctot = e["channels"]
            if len(parts) == 1 and parts[0][2] == 0 and parts[0][1] == ctot:
                acc(parts[0][0], d)
            else:
                c0 = 0
                for base, cj, up in parts:
                    hs = h >> up
                    first = base not in dval
                    if first:
                        dval[base] = torch.empty(B, cj // 16, hs, hs, 16, device=dev, dtype=torch.bfloat16)
                    check(L.ay_slice_accumulate_bf16(ptr(d), ptr(dval[base]), B, cj, ctot, c0, h, h, up, 0 if first else 1, st),
                          "ay_slice_accumulate_bf16")
                    c0 += cj
            continue
        if t == "upsample":
            if i in dval:
                d = dval.pop(i)
                c, h = e["channels"], size_of(i)
                first = e["src"] not in dval
                if first:
                    dval[e["src"]] = torch.empty(B, c // 16, h // 2, h // 2, 16, device=dev, dtype=torch.bfloat16)
                check(L.ay_slice_accumulate_bf16(ptr(d), ptr(dval[e["src"]]), B, c, c, 0, h, h, 1, 0 if first else 1, st), "ay_slice_accumulate_bf16")
            continue
        if t == "shortcut":
            continue  # handled with the fused convolution at i-1 (its gradient stays in dval[i])
        if t != "convolutional":
            continue
        rec = stt.conv[i]
        m = model.module_list[i]
        conv = m[0]
        d = rec["desc"]
        cout, cin, k = e["cout"], e["cin"], e["k"]
        hin, hout = d.hin, d.hout
        if rec["kind"] == "head":
            dh = stt.dhead.get(i + 1)
            if dh is None:
                continue
            db = torch.empty(cout, device=dev, dtype=torch.float32)
            check(L.ay_bias_grad_f32(ptr(dh), ptr(db), B, cout, hout * hout, st), "ay_bias_grad_f32")
            grads[conv.bias] = db
            dz = torch.empty(B, rec["cpad"] // 16, hout, hout, 16, device=dev, dtype=torch.bfloat16)
            check(L.ay_nchw_f32_to_blocked_bf16(ptr(dh), ptr(dz), B, cout, hout, hout, st), "ay_nchw_f32_to_blocked_bf16")
        else:
            if rec.get("fused"):
                if i + 1 not in dval:
                    continue
                dy = dval.pop(i + 1)
                acc(graph[i + 1]["b"], dy)      # the shortcut's skip operand gets the same gradient
            else:
                if i not in dval:
                    continue
                dy = dval.pop(i)
            bn = m[1]
            dz = torch.empty_like(rec["z"])
            dg = torch.empty(cout, device=dev, dtype=torch.float32)
            db = torch.empty(cout, device=dev, dtype=torch.float32)
            check(L.ay_bn_train_bwd_bf16(ptr(dy), ptr(rec["z"]), ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(rec["mean"]), ptr(rec["invstd"]),
                                         int(e["leaky"]), ptr(dz), ptr(dg), ptr(db), ptr(rec["keep"][1]), B, cout, hout, hout, st), "ay_bn_train_bwd_bf16")
            grads[bn.weight], grads[bn.bias] = dg, db
        if getattr(model, "_dbg_keep_dz", None) is not None and i in model._dbg_keep_dz:
            model._dbg_keep_dz[i] = dz
        # ---- weight gradient (matrix cores, K = pixels)
        if rec.get("stem"):
            d_w = ConvDesc(B, cin, cout, hin, hin, hout, hout, k, e["stride"], 0, 0, rec["cpad"])  # cin = 3: rows ci >= 3 of the plane are skipped
        else:
            d_w = d
        dw = torch.empty_like(rec["w"])
        check(L.ay_conv_wgrad_bf16(C.byref(d_w), ptr(rec["x"]), ptr(dz), ptr(dw), st), "ay_conv_wgrad_bf16")
        grads[conv.weight] = dw
        # ---- data gradient: the forward kernel on flipped / transposed filters
        j = rec["src"]
        if j < 0:
            continue
        cin_pad = _pad(cin, 32)
        kin = rec["cpad"]                       # channels of dz's planes (>= cout, multiple of 32)
        packed = torch.empty((kin // 16) * k * k * 2 * cin_pad * 8 * 2, device=dev, dtype=torch.uint8)
        if kin != _pad(cout, 16):               # zero rows for the planes between ceil16(cout) and kin
            packed.zero_()
        check(L.ay_pack_dgrad_weights_bf16(ptr(rec["w"]), ptr(packed), cout, cin, cin_pad, k, st), "ay_pack_dgrad_weights_bf16")
        src_dz = dz
        if e["stride"] == 2:
            up = torch.empty(B, kin // 16, hin, hin, 16, device=dev, dtype=torch.bfloat16)
            check(L.ay_zero_insert_bf16(ptr(dz), ptr(up), B, kin, hout, hout, hin, hin, st), "ay_zero_insert_bf16")
            src_dz = up
        ones, zeros = model._unit(cin_pad, dev)
        dd = ConvDesc(B, kin, cin, hin, hin, hin, hin, k, 1, 0, 0, cin_pad)
        first = j not in dval
        if first:
            dval[j] = torch.empty(B, cin_pad // 16, hin, hin, 16, device=dev, dtype=torch.bfloat16)
        check(L.ay_conv_fwd_bf16(C.byref(dd), ptr(src_dz), ptr(packed), ptr(ones), ptr(zeros), None if first else ptr(dval[j]), ptr(dval[j]), st),
              "ay_conv_fwd_bf16(dgrad)")
    out = []
    for p in model.parameters():
        g = grads.get(p)
        if g is None:
            g = torch.zeros_like(p)
        elif grad_scale != 1.0:
            g = g * grad_scale
        out.append(g)
    return out


class TrainStepBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, targets, *params):
        loss, out, stt = train_forward_bf16(model, x, targets)
        ctx.model, ctx.stt = model, stt
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        gs = float(grad_loss.item()) if grad_loss is not None else 1.0
        grads = train_backward_bf16(ctx.model, ctx.stt, gs)
        ctx.stt = None
        return (None, None, None) + tuple(grads)
