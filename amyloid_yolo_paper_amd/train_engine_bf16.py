"""Training step on the bf16 MFMA path (``Darknet(precision="bf16")``): same contract as ``train_engine`` (fp32), but

* convolutions, their data gradients (the forward kernel on re-packed filters) and their weight gradients
  (``ay_conv_wgrad_bf16``) run on the matrix cores over blocked-bf16 tensors;
* train-mode BatchNorm is a statistics pass + an apply pass around the raw convolution output, the shortcut add fused
  into the apply pass; its backward recomputes the pre-activation from the saved raw output;
* the stem runs through the same kernels (the fp32 image becomes one zero-padded 16-channel bf16 plane); the YOLO loss
  stays on the fp32 kernel;
* parameter gradients are fp32 and are ADDED straight into ``p.grad`` by the kernels (``*_acc`` entry points): with
  ``parallel.FlatGradReducer`` those are views of one flat buffer, so a step needs neither per-layer gradient temporaries
  nor autograd's accumulation pass, and ``model._grad_ready(layer)`` (set by the reducer) can start a bucket's all-reduce
  while the backward of the shallower layers is still running (reference: ``train.py:113-119``).

Memory: everything the step touches lives in a per-(batch, size) context that persists across steps -- the saved forward
tensors (raw output z and activation y of every layer: the backward needs both), small per-layer statistics, and a
shape-keyed pool for the backward's gradient tensors, which are handed back as soon as their consumer has been issued
(one stream: issue order is execution order).  Filters are packed for the forward and the data-gradient kernels once per
optimiser step (when a parameter's version or ``parallel.WEIGHT_EPOCH`` changes), not once per forward.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr
from .train_engine import METRIC_KEYS, _ws

# AY_S2_DGRAD=0: data gradient of the stride-2 layers as a stride-1 convolution over the zero-inserted output gradient (round 1)
_S2_DGRAD = os.environ.get("AY_S2_DGRAD", "1") != "0"
# AY_STEM_DIRECT=0: the stem through the generic kernels on a zero-padded 16-channel bf16 copy of the image (round 1)
_STEM_DIRECT = os.environ.get("AY_STEM_DIRECT", "1") != "0"


def _pad(v, m):
    return (v + m - 1) // m * m


def _family(e):
    """the 3x3 stride-1 layers with a multiple of 128 filters: 76 % of the model's FLOPs"""
    return e["k"] == 3 and e["stride"] == 1 and e["cout"] % 128 == 0 and e["cin"] % 32 == 0


class _Timed:
    """bench.py --mode train: HIP event pairs (on the stream the kernels are issued to) around one kind of launch;
    model._train_prof = {"wgrad": [], "conv": []} turns it on"""

    def __init__(self, prof, kind, on):
        self.lst = prof[kind] if (prof is not None and on) else None

    def __enter__(self):
        if self.lst is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.lst is not None:
            self.e1.record()
            self.lst.append((self.e0, self.e1))


class _State:
    """what one forward leaves for its backward (tests read .conv[i]["z"], .val[i])"""

    def __init__(self):
        self.val = {}
        self.conv = {}
        self.dhead = {}
        self.route = {}
        self.B = self.S = 0
        self.ctx = None


class _Pool:
    """Device memory for the gradient tensors of the backward walk: raw byte blocks with a best-fit free list, so a block
    serves tensors of different shapes over a step (per-shape free lists held 30 GB at B=32 / 1024^2: the zero-inserted
    stride-2 gradients and the full-resolution layers each pinned blocks that were in use for a fraction of the walk).
    A freed block is reused only for a request of at least half its size; everything is issued to one stream, so a block
    may be handed out again as soon as its last consumer has been ISSUED."""

    def __init__(self, dev):
        self.dev = dev
        self.free = []      # (nbytes, tensor uint8)
        self.bytes = 0
        self.owner = {}     # data_ptr of a view -> its block

    def get(self, shape, dtype=torch.bfloat16):
        n = 1
        for v in shape:
            n *= int(v)
        need = n * torch.empty(0, dtype=dtype).element_size()
        best = None
        for k, (nb, _) in enumerate(self.free):
            if need <= nb < 2 * need + 4096 and (best is None or nb < self.free[best][0]):
                best = k
        if best is not None:
            nb, block = self.free.pop(best)
        else:
            nb = (need + 255) // 256 * 256
            block = torch.empty(nb, device=self.dev, dtype=torch.uint8)
            self.bytes += nb
        t = block[:need].view(dtype).view(*shape)
        self.owner[t.data_ptr()] = (nb, block)
        return t

    def put(self, t):
        self.free.append(self.owner.pop(t.data_ptr()))


class _Ctx:
    """persistent buffers of one (batch, size) training shape"""

    def __init__(self, dev):
        self.dev = dev
        self.buf = {}
        self.pool = _Pool(dev)
        self.packed = {}
        self.packed_sig = None

    def get(self, key, shape, dtype=torch.bfloat16, zero=False):
        t = self.buf.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(*shape, device=self.dev, dtype=dtype)
            self.buf[key] = t
        return t

    def bytes(self):
        return sum(t.numel() * t.element_size() for t in self.buf.values()) + self.pool.bytes


def _weights_signature(model):
    from . import parallel
    ps = list(model.parameters())
    return (parallel.WEIGHT_EPOCH[0], sum(p._version for p in ps), ps[0].data_ptr(), ps[-1].data_ptr())


def _pack_weights(model, ctx):
    """forward and data-gradient filter images of every convolution, once per optimiser step: ONE launch over a job table
    (ay_pack_batch_bf16) that is rebuilt only when a weight tensor moved"""
    sig = _weights_signature(model)
    if ctx.packed_sig == sig:
        return ctx.packed
    import numpy as np
    L = _lib.lib()
    st = _lib.stream_ptr()
    graph = model._graph
    convs = [(i, e, model.module_list[i][0]) for i, e in enumerate(graph) if e["type"] == "convolutional"]
    table_key = tuple(c.weight.data_ptr() for _, _, c in convs)
    if getattr(ctx, "pack_table_key", None) != table_key:
        jobs, dev = [], convs[0][2].weight.device
        for i, e, conv in convs:
            cout, cin, k = e["cout"], e["cin"], e["k"]
            w = conv.weight.detach()
            rec = ctx.packed.setdefault(i, {})
            cpad = _pad(cout, 32)
            stem_direct = e["src"] < 0 and _STEM_DIRECT and cout == 32 and cin == 3 and k == 3 and e["stride"] == 1
            if e["src"] < 0:
                # the stem's generic-kernel images (16 zero-padded input channels) are built by the single-call path below; with the
                # direct kernels only the [32][32] bf16 filter table exists
                rec.update(cpad=cpad, cin_eff=16, stem_generic=not stem_direct)
                if stem_direct:
                    rec["stem_w0f"] = ctx.get(("w0f", i), (32, 32), torch.float32, zero=True)
                    rec["stem_w0"] = ctx.get(("w0", i), (32, 32), torch.bfloat16)
                else:
                    rec["fwd"] = ctx.get(("pk", i), (L.ay_packed_weight_bytes(cpad, 16, k),), torch.uint8)
                continue
            fwd = ctx.get(("pk", i), (L.ay_packed_weight_bytes(cpad, cin, k),), torch.uint8)
            rec.update(fwd=fwd, cpad=cpad, cin_eff=cin)
            jobs.append((w.data_ptr(), fwd.data_ptr(), 0, cout, cpad, cin, 0, k, fwd.numel() // 2))
            cin_pad = _pad(cin, 32)
            if e["stride"] == 2 and _S2_DGRAD:
                dg = ctx.get(("pkd", i), (L.ay_packed_dgrad_s2_weight_bytes(cpad, cin_pad),), torch.uint8)
                jobs.append((w.data_ptr(), dg.data_ptr(), 2, cout, cpad, cin, cin_pad, 3, dg.numel() // 2))
            else:
                dg = ctx.get(("pkd", i), ((cpad // 16) * k * k * 2 * cin_pad * 8 * 2,), torch.uint8, zero=True)  # rows of the pad planes stay zero
                cp16 = _pad(cout, 16)
                jobs.append((w.data_ptr(), dg.data_ptr(), 1, cout, cp16, cin, cin_pad, k, (cp16 // 16) * k * k * 2 * cin_pad * 8))
            rec.update(dgrad=dg, cin_pad=cin_pad)
        job_dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("kind", "<i4"), ("cout", "<i4"), ("cout_pad", "<i4"), ("cin", "<i4"),
                           ("cin_pad", "<i4"), ("ksize", "<i4"), ("total", "<u8")])
        assert job_dt.itemsize == 48
        ja = np.array(jobs, dtype=job_dt)
        blk = L.ay_pack_batch_block()
        work = np.array([(j, b) for j, jb in enumerate(jobs) for b in range((jb[8] + blk - 1) // blk)], dtype=np.dtype([("job", "<u4"), ("first", "<u4")]))
        ctx.pack_jobs = torch.from_numpy(ja.view(np.uint8).copy()).to(dev)
        ctx.pack_work = torch.from_numpy(work.view(np.uint8).copy()).to(dev)
        ctx.pack_n_work = int(work.shape[0])
        ctx.pack_table_key = table_key
    check(L.ay_pack_batch_bf16(ptr(ctx.pack_jobs), ptr(ctx.pack_work), ctx.pack_n_work, st), "ay_pack_batch_bf16")
    for i, e, conv in convs:
        if e["src"] >= 0:
            continue
        rec, w = ctx.packed[i], conv.weight.detach()
        if rec.get("stem_w0") is not None:
            rec["stem_w0f"][:, :27].copy_(w.reshape(32, 27))
            rec["stem_w0"].copy_(rec["stem_w0f"])
        else:  # stem through the generic kernels: filters get zero input channels 3..15
            w16 = ctx.get(("w16", i), (e["cout"], 16, e["k"], e["k"]), torch.float32, zero=True)
            w16[:, : e["cin"]].copy_(w)
            check(L.ay_pack_conv_weights_bf16(ptr(w16), ptr(rec["fwd"]), e["cout"], rec["cpad"], 16, e["k"], st), "ay_pack_conv_weights_bf16")
    ctx.packed_sig = sig
    return ctx.packed


def _context(model, B, S, dev):
    ctxs = model.__dict__.setdefault("_train_ctx", {})
    key = (B, S, str(dev))
    if key not in ctxs:
        if len(ctxs) >= 2:  # multiscale training walks through sizes: keep the two most recent shapes
            ctxs.pop(next(iter(ctxs)))
        ctxs[key] = _Ctx(dev)
    return ctxs[key]


def train_forward_bf16(model, x, targets):
    L = _lib.lib()
    st = _lib.stream_ptr()
    dev = torch.device("cuda", torch.cuda.current_device())
    x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
    B, _, S, S2 = x.shape
    assert S == S2 and S % 32 == 0
    assert model.training, "the bf16 training path implements train-mode BatchNorm only"
    graph = model._graph
    ctx = _context(model, B, S, dev)
    packed = _pack_weights(model, ctx)
    stt = _State()
    stt.B, stt.S, stt.ctx = B, S, ctx
    val = stt.val
    Ccls = model.yolo_layers[0].num_classes
    N = model.num_boxes(S)
    out = ctx.get("out", (B, N, 5 + Ccls), torch.float32)
    tg = None if targets is None else targets.detach().to(device=dev, dtype=torch.float32).contiguous()

    def size_of(i):
        return S >> graph[i]["log2_down"] if i >= 0 else S

    def blocked(key, c, h, dtype=torch.bfloat16, pad=16):
        return ctx.get(key, (B, _pad(c, pad) // 16, h, h, 16), dtype)

    def resolve(i):
        v = val[i]
        if isinstance(v, tuple):
            src = resolve(v[1])
            c, h = graph[i]["channels"], size_of(i)
            o = blocked(("up", i), c, h)
            check(L.ay_concat_upsample_bf16(ptr(src), c, 1, None, 0, ptr(o), B, h, h, st), "ay_concat_upsample_bf16")
            val[i] = o
            return o
        return v

    # fp64 statistics workspaces of all BatchNorm layers, forward pass [0] and backward pass [1], as slices of ONE tensor cleared by
    # one memset per step (the library's plain entry points clear theirs per call: 144 fill launches per step)
    bn_layers = [j for j, g in enumerate(graph) if g["type"] == "convolutional" and g["bn"]]
    off, bn_off = 0, {}
    for j in bn_layers:
        bn_off[j] = off
        off += 4 * graph[j]["cout"]
    bn_ws_flat = ctx.get("bn_ws_flat", (max(off, 1),), torch.float64)
    bn_ws_flat.zero_()
    ctx.bn_ws_clean = True    # the backward halves are zero: ONE backward pass may accumulate into them (train_backward_bf16)

    def bn_ws(j, backward):
        o = bn_off[j] + (2 * graph[j]["cout"] if backward else 0)
        return bn_ws_flat[o:o + 2 * graph[j]["cout"]]

    row = 0
    sums_all = []
    nbt = []   # BatchNorm modules that ran: their num_batches_tracked counters are views of one flat tensor, bumped once per forward
    prof = getattr(model, "_train_prof", None)
    for i, e in enumerate(graph):
        t = e["type"]
        if t == "convolutional":
            m = model.module_list[i]
            conv = m[0]
            hin, hout = size_of(e["src"]), size_of(i)
            cout, cin, k = e["cout"], e["cin"], e["k"]
            pk = packed[i]
            cpad, cin_eff = pk["cpad"], pk["cin_eff"]
            stem_direct = e["src"] < 0 and "stem_w0" in pk and hin % 4 == 0
            if stem_direct:
                src = x   # ---- stem straight from the fp32 image: forward and filter gradient read it once (ay_stem_train.hip)
            elif e["src"] < 0:
                # ---- stem: the fp32 image becomes ONE zero-padded 16-channel bf16 plane and the layer runs through the
                # same MFMA kernels as every other layer
                assert e["bn"] and cin <= 16
                src = blocked("xb", 16, hin)
                check(L.ay_nchw_f32_to_blocked_bf16(ptr(x), ptr(src), B, cin, hin, hin, st), "ay_nchw_f32_to_blocked_bf16")
            else:
                src = resolve(e["src"])
            is_head = not e["bn"]
            ones, zeros = model._unit(cpad, dev)
            d = ConvDesc(B, cin_eff, cout, hin, hin, hout, hout, k, e["stride"], 0, int(is_head), cpad)
            rec = dict(kind="head" if is_head else "bn", x=src, desc=d, src=e["src"], cpad=cpad, stem=e["src"] < 0, stem_direct=stem_direct)
            if is_head:
                shift = ctx.get(("shift", i), (cpad,), torch.float32, zero=True)
                shift[:cout].copy_(conv.bias.detach())
                zb = blocked(("zb", i), cout, hout, torch.float32, 32)
                check(L.ay_conv_fwd_bf16(C.byref(d), ptr(src), ptr(pk["fwd"]), ptr(ones), ptr(shift), None, ptr(zb), st), "ay_conv_fwd_bf16")
                head = ctx.get(("head", i), (B, cout, hout, hout), torch.float32)
                check(L.ay_blocked_f32_to_nchw_f32(ptr(zb), ptr(head), B, cout, hout, hout, st), "ay_blocked_f32_to_nchw_f32")
                stt.conv[i] = rec
                val[i] = head
                continue
            bn = m[1]
            assert cout % 32 == 0, f"layer {i}: the bf16 training path needs BN layers with a multiple of 32 filters (got {cout})"
            z = blocked(("z", i), cout, hout, pad=32)
            stats_done = False
            if stem_direct:
                # forward + the layer's BatchNorm batch statistics in one kernel (the sums land in the layer's fp64 workspace)
                ws = bn_ws(i, False)
                nws = L.ay_stem_train_stats_workspace_bytes()
                sws = ctx.buf.get("stem_stats_ws")
                if sws is None or sws.numel() < nws:
                    sws = ctx.buf["stem_stats_ws"] = torch.empty(nws, device=dev, dtype=torch.uint8)
                check(L.ay_stem_train_fwd_stats_bf16(ptr(x), ptr(pk["stem_w0"]), ptr(z), ptr(ws), ptr(sws), sws.numel(), B, hin, hin, st),
                      "ay_stem_train_fwd_stats_bf16")
                stats_done = True
            else:
                with _Timed(prof, "conv", _family(e)):
                    check(L.ay_conv_fwd_bf16(C.byref(d), ptr(src), ptr(pk["fwd"]), ptr(ones), ptr(zeros), None, ptr(z), st), "ay_conv_fwd_bf16")
            fuse = e["fuse_into_shortcut"]
            skip = resolve(graph[i + 1]["b"]) if fuse else None
            y = blocked(("y", i), cout, hout, pad=32)
            mean = ctx.get(("mean", i), (cout,), torch.float32)
            invstd = ctx.get(("invstd", i), (cout,), torch.float32)
            ws = bn_ws(i, False)
            check((L.ay_bn_train_apply_bf16 if stats_done else L.ay_bn_train_fwd_bf16_zeroed_ws)(
                ptr(z), ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(bn.running_mean), ptr(bn.running_var), C.c_float(bn.momentum),
                C.c_float(bn.eps), int(e["leaky"]), ptr(skip), ptr(y), ptr(mean), ptr(invstd), ptr(ws), B, cout, hout, hout, st),
                "ay_bn_train_fwd_bf16")
            nbt.append(bn)
            rec.update(z=z, mean=mean, invstd=invstd, fused=fuse, ws=bn_ws(i, True))
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
                o = blocked(("route", i), e["channels"], h)
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
                dhead = ctx.get(("dhead", i), tuple(head.shape), torch.float32)
                sums = ctx.get(("sums", i), (16,), torch.float32)
                nb = L.ay_yolo_loss_workspace_bytes(B, y.num_anchors, y.num_classes, G)
                ws = _ws(model, nb, dev)
                check((L.ay_yolo_loss_giou_fwd_bwd if getattr(model, 'box_loss', 'mse') == 'giou' else L.ay_yolo_loss_fwd_bwd)(ptr(head), ptr(tg), tg.shape[0], B, y.num_anchors, y.num_classes, G, S, anchors,
                                             C.c_float(y.ignore_thres), C.c_float(1.0), ptr(dhead), ptr(sums), ptr(ws), ws.numel(), st),
                      "ay_yolo_loss_fwd_bwd")
                stt.dhead[i] = dhead
                sums_all.append((y, sums, G))
            val[i] = head
    if nbt:
        # num_batches_tracked += 1 for every BatchNorm that ran (models.py:43 semantics of nn.BatchNorm2d in train mode): the counters
        # are re-pointed once to views of one flat int64 tensor, so a step costs one add instead of 72 launches
        flat = getattr(model, "_nbt_flat", None)
        if (flat is None or flat.numel() != len(nbt) or flat.device != dev or nbt[0].num_batches_tracked.data_ptr() != flat[0].data_ptr()
                or nbt[-1].num_batches_tracked.data_ptr() != flat[-1].data_ptr()):
            flat = torch.stack([b.num_batches_tracked.detach().to(dev) for b in nbt])
            for k_, b in enumerate(nbt):
                b._buffers["num_batches_tracked"] = flat[k_]
            model._nbt_flat = flat
        flat.add_(1)
    loss = None
    if tg is not None:
        allsums = torch.stack([s for _, s, _ in sums_all])
        n_obj, n_noobj = allsums[:, 7], allsums[:, 8]
        lx, ly, lw, lh = (allsums[:, k] / n_obj for k in range(4))
        lconf = allsums[:, 4] / n_obj + 100.0 * allsums[:, 5] / n_noobj
        lcls = allsums[:, 6] / (n_obj * Ccls)
        per_layer = lx + ly + lw + lh + lconf + lcls
        loss = per_layer.sum()
        if getattr(model, "collect_metrics", True):   # one host sync per step (the reference: 39, models.py:205-220)
            h = torch.stack([per_layer, lx, ly, lw, lh, lconf, lcls, 100.0 * allsums[:, 9] / n_obj, allsums[:, 13] / (n_obj + 1e-16),
                             allsums[:, 14] / (n_obj + 1e-16), allsums[:, 13] / (allsums[:, 12] + 1e-16), allsums[:, 10] / n_obj,
                             allsums[:, 11] / n_noobj], 1).cpu().numpy()
            for li, (y, _, G) in enumerate(sums_all):
                y.metrics = {k: float(h[li, j]) for j, k in enumerate(METRIC_KEYS[:-1])}
                y.metrics["grid_size"] = G
    return loss, out, stt


def train_backward_bf16(model, stt, grad_scale=None):
    """walks the plan in reverse; every parameter gradient is ADDED into ``p.grad`` (created as zeros where missing)"""
    L = _lib.lib()
    st = _lib.stream_ptr()
    graph = model._graph
    B, S = stt.B, stt.S
    ctx = stt.ctx
    pool = ctx.pool
    packed = ctx.packed
    dev = ctx.dev
    dval = {}
    # ay_bn_train_bwd_bf16_acc_zeroed_ws ADDS its fp64 sums onto the workspace and trusts it to be zero.  The forward cleared it; a
    # second backward over the same state (or one without a fresh forward of this (B, S) context) would add onto dirty sums and give
    # silently wrong dgamma / dbeta / dz: clear it here then (the forward halves are not read again: mean / invstd are saved per layer)
    if not getattr(ctx, "bn_ws_clean", False):
        ctx.buf["bn_ws_flat"].zero_()
    ctx.bn_ws_clean = False
    for p in model.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    hook = getattr(model, "_grad_ready", None)
    keep_dz = getattr(model, "_dbg_keep_dz", None)
    prof = getattr(model, "_train_prof", None)

    def size_of(i):
        return S >> graph[i]["log2_down"] if i >= 0 else S

    def acc(j, t, own):
        """accumulate the blocked-bf16 gradient t into layer j's output gradient; `own`: t came from the pool and may be kept"""
        if j < 0:
            if own:
                pool.put(t)
            return
        if j not in dval:
            if own:
                dval[j] = t
            else:
                c = pool.get(t.shape)
                c.copy_(t)
                dval[j] = c
        else:
            check(L.ay_accumulate_bf16(ptr(dval[j]), ptr(t), t.numel(), st), "ay_accumulate_bf16")
            if own:
                pool.put(t)

    if grad_scale is not None:   # d(loss)/d(loss) handed down by autograd: a device scalar, applied without a host sync
        for dh in stt.dhead.values():
            dh.mul_(grad_scale)
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
                acc(parts[0][0], d, True)
            else:
                c0 = 0
                for base, cj, up in parts:
                    hs = h >> up
                    first = base not in dval
                    if first:
                        dval[base] = pool.get((B, cj // 16, hs, hs, 16))
                    check(L.ay_slice_accumulate_bf16(ptr(d), ptr(dval[base]), B, cj, ctot, c0, h, h, up, 0 if first else 1, st),
                          "ay_slice_accumulate_bf16")
                    c0 += cj
                pool.put(d)
            continue
        if t == "upsample":
            if i in dval:
                d = dval.pop(i)
                c, h = e["channels"], size_of(i)
                first = e["src"] not in dval
                if first:
                    dval[e["src"]] = pool.get((B, c // 16, h // 2, h // 2, 16))
                check(L.ay_slice_accumulate_bf16(ptr(d), ptr(dval[e["src"]]), B, c, c, 0, h, h, 1, 0 if first else 1, st), "ay_slice_accumulate_bf16")
                pool.put(d)
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
            check(L.ay_bias_grad_f32_acc(ptr(dh), ptr(conv.bias.grad), 1, B, cout, hout * hout, st), "ay_bias_grad_f32")
            dz = pool.get((B, rec["cpad"] // 16, hout, hout, 16))
            if rec["cpad"] != _pad(cout, 16):  # planes between ceil16(cout) and cpad are not written by the converter
                dz[:, _pad(cout, 16) // 16:].zero_()
            check(L.ay_nchw_f32_to_blocked_bf16(ptr(dh), ptr(dz), B, cout, hout, hout, st), "ay_nchw_f32_to_blocked_bf16")
        else:
            if rec.get("fused"):
                if i + 1 not in dval:
                    continue
                dy = dval.pop(i + 1)
            else:
                if i not in dval:
                    continue
                dy = dval.pop(i)
            bn = m[1]
            dz = pool.get(tuple(rec["z"].shape))
            check(L.ay_bn_train_bwd_bf16_acc_zeroed_ws(ptr(dy), ptr(rec["z"]), ptr(bn.weight.detach()), ptr(bn.bias.detach()), ptr(rec["mean"]), ptr(rec["invstd"]),
                                             int(e["leaky"]), ptr(dz), ptr(bn.weight.grad), ptr(bn.bias.grad), ptr(rec["ws"]), 1, B, cout, hout, hout, st),
                  "ay_bn_train_bwd_bf16")
            if rec.get("fused"):
                # the shortcut's skip operand gets the same gradient: dy itself moves on (the BN backward above only read it, and
                # whatever accumulates into it later is issued later on the same stream) -- no copy
                acc(graph[i + 1]["b"], dy, True)
            else:
                pool.put(dy)
        if keep_dz is not None and i in keep_dz:
            keep_dz[i] = dz.clone()
        # ---- weight gradient (matrix cores, K = pixels), added into conv.weight.grad
        if rec.get("stem_direct"):
            nws = L.ay_stem_train_wgrad_workspace_bytes()
            wws = ctx.buf.get("wgrad_ws")
            if wws is None or wws.numel() < nws:
                wws = ctx.buf["wgrad_ws"] = torch.empty(max(nws, 1 << 20), device=dev, dtype=torch.uint8)
            check(L.ay_stem_train_wgrad_bf16(ptr(rec["x"]), ptr(dz), ptr(conv.weight.grad), 1, ptr(wws), wws.numel(), B, hin, hin, st),
                  "ay_stem_train_wgrad_bf16")
            if hook is not None:
                hook(i)
            pool.put(dz)
            continue
        if rec.get("stem"):
            d_w = ConvDesc(B, cin, cout, hin, hin, hout, hout, k, e["stride"], 0, 0, rec["cpad"])  # cin = 3: rows ci >= 3 of the plane are skipped
        else:
            d_w = d
        # split-K partial filters go to slabs in a workspace and are summed in a fixed order (reproducible; no fp32 atomics)
        nws = L.ay_conv_wgrad_workspace_bytes(C.byref(d_w))
        wws = ctx.buf.get("wgrad_ws")
        if wws is None or wws.numel() < nws:
            wws = ctx.buf["wgrad_ws"] = torch.empty(max(nws, 1 << 20), device=dev, dtype=torch.uint8)
        with _Timed(prof, "wgrad", _family(e)):
            check(L.ay_conv_wgrad_bf16_ws(C.byref(d_w), ptr(rec["x"]), ptr(dz), ptr(conv.weight.grad), 1, ptr(wws), wws.numel(), st), "ay_conv_wgrad_bf16")
        if hook is not None:
            hook(i)
        # ---- data gradient: the forward kernel on flipped / transposed filters
        j = rec["src"]
        if j < 0:
            pool.put(dz)
            continue
        pk = packed[i]
        cin_pad = pk["cin_pad"]
        kin = rec["cpad"]                       # channels of dz's planes (>= cout, multiple of 32)
        src_dz = dz
        if e["stride"] == 2 and _S2_DGRAD:
            # four stride-1 sub-convolutions of dz with a 2x2 window, one per parity class of the input pixel
            ones, zeros = model._unit(cin_pad, dev)
            first = j not in dval
            if first:
                dval[j] = pool.get((B, cin_pad // 16, hin, hin, 16))
            check(L.ay_conv_dgrad_s2_bf16(C.byref(d), ptr(dz), ptr(pk["dgrad"]), ptr(ones), ptr(zeros), None if first else ptr(dval[j]), ptr(dval[j]),
                                          cin_pad, st), "ay_conv_dgrad_s2_bf16")
            pool.put(dz)
            continue
        if e["stride"] == 2:
            up = pool.get((B, kin // 16, hin, hin, 16))
            check(L.ay_zero_insert_bf16(ptr(dz), ptr(up), B, kin, hout, hout, hin, hin, st), "ay_zero_insert_bf16")
            src_dz = up
        ones, zeros = model._unit(cin_pad, dev)
        dd = ConvDesc(B, kin, cin, hin, hin, hin, hin, k, 1, 0, 0, cin_pad)
        first = j not in dval
        if first:
            dval[j] = pool.get((B, cin_pad // 16, hin, hin, 16))
        # (timed with the family where the data gradient is itself a 3x3 s1 convolution onto a multiple of 128 channels)
        with _Timed(prof, "conv", _family(e) and e["stride"] == 1 and cin % 128 == 0):
            check(L.ay_conv_fwd_bf16(C.byref(dd), ptr(src_dz), ptr(pk["dgrad"]), ptr(ones), ptr(zeros), None if first else ptr(dval[j]), ptr(dval[j]), st),
                  "ay_conv_fwd_bf16(dgrad)")
        pool.put(dz)
        if src_dz is not dz:
            pool.put(src_dz)
    for t in dval.values():
        pool.put(t)
    if hook is not None:
        hook(-1)


class TrainStepBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, targets, *params):
        loss, out, stt = train_forward_bf16(model, x, targets)
        ctx.model, ctx.stt = model, stt
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, grad_loss, _grad_out):
        train_backward_bf16(ctx.model, ctx.stt, grad_loss)   # adds into every p.grad itself
        n = len(list(ctx.model.parameters()))
        ctx.stt = None
        return (None, None, None) + (None,) * n
