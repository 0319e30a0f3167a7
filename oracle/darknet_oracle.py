"""PyTorch-CPU fp32 restatement of the reference's Darknet graph (TEST INFRASTRUCTURE ONLY).

Follows ``models.py:16-83`` (cfg blocks -> layers), ``models.py:237-255``
(interpreter), ``models.py:127-222`` (YOLO decode + loss) and
``models.py:257-308`` (Darknet ``.weights`` layout) of the reference.  The
convolution / batch-norm arithmetic itself is PyTorch ATen on the CPU, exactly
what the reference dispatches to (SURVEY.md §8c "third-party arithmetic").

``mode="bf16"`` emulates the numeric contract of the HIP bf16 path so layer
outputs can be compared tightly: conv operands rounded to bfloat16, fp32
accumulation, fp32 BN-affine + LeakyReLU (+ residual) epilogue, ONE rounding to
bfloat16 per stored activation; the three linear heads (fp32 out) are not rounded.
The stem takes the fp32 image: with ``stem_bf16=True`` (the fused stem kernel, default) image and stem filters are
rounded to bfloat16 like every other operand, with ``stem_bf16=False`` (the separate fp32 stem kernel) they are not.

``mode="fp16"`` is the same contract with IEEE half in the place of bfloat16 (``precision="fp16"`` of the HIP path: the
same rounding points -- operands, one rounding per stored activation -- with an 11-bit significand instead of 8).

``mode="bf16_train"`` is the numeric contract of the HIP bf16 TRAINING step (``train_engine_bf16.py``): as ``bf16``, and
in addition the raw convolution output z of a BN layer is rounded to bfloat16 before the batch statistics are taken (the
training path stores z and normalises in a second pass).  Every rounding is a straight-through estimator
(``x + (bf16(x) - x).detach()``), so ``loss.backward()`` yields fp32 autograd gradients evaluated on the bf16 forward
activations -- the same LeakyReLU masks and BN statistics the HIP step sees; what the HIP step adds on top is one bf16
rounding per stored activation gradient.  The mode only inserts roundings into the code path that the reference's own
training fixtures pin in ``mode="fp32"`` (tests/golden/train_*.npz).

A bf16 forward is chaotic at the one-ulp level: an fp32 sum that lands next to a bf16 rounding boundary goes either way with
the summation order, and each such flip changes the roundings of many outputs of the next layer (measured on the 75-conv
network in train mode: 5e-6 of the elements differ after layer 0, 2e-3 after layer 2, half of them after layer 9; 1-9 % relative
L2 in the deep layers).  Two exact evaluations of the same contract therefore differ; ``conv_f64=True`` gives a second one
(exact convolution sums) so that tests can bound "HIP vs oracle" by "oracle vs oracle".
"""
import numpy as np
import torch
import torch.nn.functional as F


def parse_cfg(path):
    """utils/parse_config.py:3-21 (own restatement; values stay strings)."""
    blocks = []
    for raw in open(path).read().split("\n"):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("["):
            blocks.append({"type": line[1:-1].rstrip()})
            if blocks[-1]["type"] == "convolutional":
                blocks[-1]["batch_normalize"] = 0
        else:
            k, v = line.split("=")
            blocks[-1][k.rstrip()] = v.strip()
    return blocks


def giou_cxcywh(b1, b2):
    """GIoU of (cx,cy,w,h) boxes [n,4] x [n,4] with torch ops (differentiable); same formula as boxes_oracle.bbox_giou."""
    x1, x2, y1, y2 = b1[:, 0] - b1[:, 2] / 2, b1[:, 0] + b1[:, 2] / 2, b1[:, 1] - b1[:, 3] / 2, b1[:, 1] + b1[:, 3] / 2
    X1, X2, Y1, Y2 = b2[:, 0] - b2[:, 2] / 2, b2[:, 0] + b2[:, 2] / 2, b2[:, 1] - b2[:, 3] / 2, b2[:, 1] + b2[:, 3] / 2
    iw = (torch.min(x2, X2) - torch.max(x1, X1)).clamp(min=0)
    ih = (torch.min(y2, Y2) - torch.max(y1, Y1)).clamp(min=0)
    inter = iw * ih
    union = (x2 - x1) * (y2 - y1) + (X2 - X1) * (Y2 - Y1) - inter + 1e-16
    hull = (torch.max(x2, X2) - torch.min(x1, X1)) * (torch.max(y2, Y2) - torch.min(y1, Y1)) + 1e-16
    return inter / union - (hull - union) / hull


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _fp16(t):
    return t.to(torch.float16).to(torch.float32)


def _ulp(mode):
    """relative size of one rounding step of the storage type of `mode`"""
    return 2.0 ** -10 if mode == "fp16" else 2.0 ** -7


def _ulp_report(own, forced, f32=False):
    """(fraction of elements further than one bf16 ulp -- fp32 values: 1e-5 relative -- from `forced`, relative L2)"""
    err = (own - forced).abs()
    tol = forced.abs() * (1e-5 if f32 else 2.0 ** -7) + 1e-6
    return float((err > tol).float().mean()), float(err.norm() / (forced.norm() + 1e-30))


def _ste(t):
    """bf16 rounding with a straight-through gradient"""
    return t + (_bf16(t) - t).detach()


class OracleDarknet:
    def __init__(self, cfg_path):
        defs = parse_cfg(cfg_path)
        self.hyper = defs[0]
        self.defs = defs[1:]
        self.params = {}  # layer idx -> dict of torch tensors
        filters = [int(self.hyper["channels"])]
        for i, d in enumerate(self.defs):
            t = d["type"]
            if t == "convolutional":
                cin, cout, k = filters[-1], int(d["filters"]), int(d["size"])
                p = {"weight": torch.zeros(cout, cin, k, k)}
                if int(d["batch_normalize"]):
                    p.update(gamma=torch.ones(cout), beta=torch.zeros(cout), mean=torch.zeros(cout), var=torch.ones(cout))
                else:
                    p["bias"] = torch.zeros(cout)
                self.params[i] = p
                f = cout
            elif t == "route":
                f = sum(filters[1:][int(x)] for x in d["layers"].split(","))
            elif t == "shortcut":
                f = filters[1:][int(d["from"])]
            else:
                f = filters[-1]
            filters.append(f)
        self.seen = 0
        self.metrics = []
        self.keep_z = None     # {} -> raw convolution outputs of the BN layers, with .grad after backward (diagnosis)
        self.conv_f64 = False  # convolution sums in float64 (rounded to fp32 once) instead of ATen's fp32 order
        self.box_loss = "mse"  # "giou": 1 - GIoU box term (no reference counterpart: parity unpinned, checked against autograd)

    # ---- weights (models.py:257-308) -------------------------------------------------
    def load_darknet_weights(self, path):
        with open(path, "rb") as fh:
            header = np.fromfile(fh, dtype=np.int32, count=5)
            w = np.fromfile(fh, dtype=np.float32)
        self.seen = int(header[3])
        ptr = 0
        for i, d in enumerate(self.defs):
            if d["type"] != "convolutional":
                continue
            p = self.params[i]
            names = ("beta", "gamma", "mean", "var") if int(d["batch_normalize"]) else ("bias",)
            for n in names + ("weight",):
                cnt = p[n].numel()
                p[n] = torch.from_numpy(w[ptr:ptr + cnt].copy()).view_as(p[n])
                ptr += cnt
        assert ptr == w.size, (ptr, w.size)

    def set_params(self, params):
        for i, p in params.items():
            for k, v in p.items():
                self.params[i][k] = torch.from_numpy(np.ascontiguousarray(v, np.float32)).clone()

    def require_grad(self):
        for p in self.params.values():
            for k in ("weight", "gamma", "beta", "bias"):
                if k in p:
                    p[k] = p[k].detach().clone().requires_grad_(True)

    # ---- forward ----------------------------------------------------------------------
    def _conv_block(self, i, d, x, mode, train_bn, stem_bf16=True, forced_z=None):
        p = self.params[i]
        k, s = int(d["size"]), int(d["stride"])
        w = p["weight"]
        first = x.shape[1] == int(self.hyper["channels"]) and i == 0
        if mode in ("bf16", "fp16") and (not first or stem_bf16):
            r16 = _bf16 if mode == "bf16" else _fp16
            w = r16(w)
            if first:
                x = r16(x)
        if mode == "bf16_train":
            w = _ste(w)
            if first:
                x = _bf16(x)
        if self.conv_f64:  # exact sums, rounded to fp32 once: a second valid evaluation of the same contract (summation order)
            y = F.conv2d(x.double(), w.double(), None, stride=s, padding=(k - 1) // 2).float()
        else:
            y = F.conv2d(x, w, None, stride=s, padding=(k - 1) // 2)
        if mode == "bf16_train" and int(d["batch_normalize"]):
            if forced_z is not None:  # teacher forcing: the value the checked implementation stored, this layer's Jacobian
                self.forced_err[("z", i)] = _ulp_report(_bf16(y.detach()), forced_z)
                y = forced_z + (y - y.detach())
            else:
                y = _ste(y)  # the stored raw output z
        if self.keep_z is not None and int(d["batch_normalize"]):  # tests: gradient w.r.t. the raw convolution output
            y.retain_grad()
            self.keep_z[i] = y
        if int(d["batch_normalize"]):
            if train_bn:
                # models.py:43 -- PyTorch momentum 0.9 semantics
                y = F.batch_norm(y, p["mean"], p["var"], p["gamma"], p["beta"], True, 0.9, 1e-5)
            else:
                scale = p["gamma"] / torch.sqrt(p["var"] + 1e-5)
                shift = p["beta"] - p["mean"] * scale
                y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        else:
            y = y + p["bias"].view(1, -1, 1, 1)
        if d["activation"] == "leaky":
            y = F.leaky_relu(y, 0.1)
        return y

    def forward(self, x, targets=None, mode="fp32", train_bn=False, collect=False, stem_bf16=True, forced=None, layer_modes=None):
        """x [B,3,S,S] float32 tensor.  Returns outputs [B,N,5+C] (and loss if targets).

        ``layer_modes`` (inference experiments, oracle/parity_sweep.py): callable layer index -> "bf16" | "fp16" | "fp32", the storage
        type of that layer's filters and of its stored output (a shortcut's sum is stored in the type of the shortcut layer's index);
        overrides ``mode`` per layer.

        ``forced`` (``mode="bf16_train"`` only) = {"z": {layer: NCHW fp32}, "y": {layer: NCHW fp32}}: stored raw convolution
        outputs and stored layer outputs of the implementation under test.  Each layer is then evaluated on the forced inputs,
        its own result is compared with the forced value (``self.forced_err``: fraction of elements more than one bf16 ulp
        off, relative L2) and REPLACED by it, keeping this layer's Jacobian: ``forced + (f(x) - f(x).detach())``.  The
        backward pass is then fp32 autograd at exactly the forward point of the implementation under test -- same LeakyReLU
        masks, same batch statistics -- which removes the one-ulp chaos described in the module docstring from a
        gradient comparison."""
        self.forced_err = {}
        fz = (forced or {}).get("z", {})
        fy = (forced or {}).get("y", {})

        def force(i, xf, rounded):
            if i in fy:
                is_f32 = rounded is xf
                self.forced_err[("y", i)] = _ulp_report(rounded.detach(), fy[i], f32=is_f32)
                return fy[i] + (xf - xf.detach())
            return rounded

        img_dim = x.shape[2]
        outs_r, outs_f = [], []  # stored (rounded) and unrounded fp32 value of every layer output
        yolo_out, loss = [], 0
        self.metrics = []
        rnd = _bf16 if mode == "bf16" else _fp16 if mode == "fp16" else _ste if mode == "bf16_train" else (lambda t: t)
        mode0 = mode
        for i, d in enumerate(self.defs):
            t = d["type"]
            if layer_modes is not None:
                mode = layer_modes(i)
                rnd = _bf16 if mode == "bf16" else _fp16 if mode == "fp16" else (lambda t: t)
            if t == "convolutional":
                xf = self._conv_block(i, d, x, mode, train_bn, stem_bf16, fz.get(i))
                is_head = not int(d["batch_normalize"])
                x = force(i, xf, xf if is_head else rnd(xf))
            elif t == "upsample":
                xf = x = F.interpolate(x, scale_factor=int(d["stride"]), mode="nearest")
            elif t == "route":
                xf = x = torch.cat([outs_r[int(j)] for j in d["layers"].split(",")], 1)
            elif t == "shortcut":
                xf = outs_f[-1] + outs_r[int(d["from"])]
                x = force(i, xf, rnd(xf))
            elif t == "yolo":
                x, layer_loss = self._yolo(d, x, targets, img_dim)
                xf = x
                loss = loss + layer_loss
                yolo_out.append(x)
            outs_r.append(x)
            outs_f.append(xf)
        self.layer_outputs = outs_r if collect else None
        out = torch.cat(yolo_out, 1).detach()
        return out if targets is None else (loss, out)

    def _yolo(self, d, x, targets, img_dim):
        """models.py:127-222."""
        mask = [int(v) for v in d["mask"].split(",")]
        a = [int(v) for v in d["anchors"].split(",")]
        anchors = [(a[2 * m], a[2 * m + 1]) for m in mask]
        C = int(d["classes"])
        B, _, G, _ = x.shape
        A = len(anchors)
        pred = x.view(B, A, C + 5, G, G).permute(0, 1, 3, 4, 2).contiguous()
        sx, sy = torch.sigmoid(pred[..., 0]), torch.sigmoid(pred[..., 1])
        w, h = pred[..., 2], pred[..., 3]
        conf, cls = torch.sigmoid(pred[..., 4]), torch.sigmoid(pred[..., 5:])
        stride = img_dim / G
        gx = torch.arange(G, dtype=torch.float32).repeat(G, 1).view(1, 1, G, G)
        gy = torch.arange(G, dtype=torch.float32).repeat(G, 1).t().view(1, 1, G, G)
        sa = torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors], dtype=torch.float32)
        boxes = torch.empty(pred[..., :4].shape)
        boxes[..., 0] = sx.detach() + gx
        boxes[..., 1] = sy.detach() + gy
        boxes[..., 2] = torch.exp(w.detach()) * sa[:, 0].view(1, A, 1, 1)
        boxes[..., 3] = torch.exp(h.detach()) * sa[:, 1].view(1, A, 1, 1)
        out = torch.cat((boxes.view(B, -1, 4) * stride, conf.view(B, -1, 1), cls.view(B, -1, C)), -1)
        if targets is None:
            return out, 0
        from . import boxes_oracle as bo
        tg = bo.build_targets(boxes.numpy(), cls.detach().numpy(), targets.numpy(), sa.numpy(), 0.5)
        iou_scores, class_mask, obj, noobj, tx, ty, tw, th, tcls, tconf = [torch.from_numpy(np.ascontiguousarray(v)) for v in tg]
        mse, bce = F.mse_loss, F.binary_cross_entropy
        if self.box_loss == "giou":
            # published GIoU on corner boxes in grid units (no +1 rule), target box of the winning target per cell
            gi = gx.expand(B, A, G, G)[obj]
            gj = gy.expand(B, A, G, G)[obj]
            aw = sa[:, 0].view(1, A, 1, 1).expand(B, A, G, G)[obj]
            ah = sa[:, 1].view(1, A, 1, 1).expand(B, A, G, G)[obj]
            pb = torch.stack((sx[obj] + gi, sy[obj] + gj, torch.exp(w[obj]) * aw, torch.exp(h[obj]) * ah), 1)
            tb = torch.stack((tx[obj] + gi, ty[obj] + gj, torch.exp(tw[obj]) * aw, torch.exp(th[obj]) * ah), 1)
            lx = (1.0 - giou_cxcywh(pb, tb)).mean()
            ly = lw = lh = torch.zeros(())
        else:
            lx, ly = mse(sx[obj], tx[obj]), mse(sy[obj], ty[obj])
            lw, lh = mse(w[obj], tw[obj]), mse(h[obj], th[obj])
        lconf_obj = bce(conf[obj], tconf[obj])
        lconf_noobj = bce(conf[noobj], tconf[noobj])
        lconf = 1 * lconf_obj + 100 * lconf_noobj
        lcls = bce(cls[obj], tcls[obj])
        total = lx + ly + lw + lh + lconf + lcls
        conf50 = (conf > 0.5).float()
        iou50, iou75 = (iou_scores > 0.5).float(), (iou_scores > 0.75).float()
        det = conf50 * class_mask
        self.metrics.append({
            "loss": total.item(), "x": lx.item(), "y": ly.item(), "w": lw.item(), "h": lh.item(),
            "conf": lconf.item(), "cls": lcls.item(), "cls_acc": (100 * class_mask[obj].mean()).item(),
            "recall50": (torch.sum(iou50 * det) / (obj.sum() + 1e-16)).item(),
            "recall75": (torch.sum(iou75 * det) / (obj.sum() + 1e-16)).item(),
            "precision": (torch.sum(iou50 * det) / (conf50.sum() + 1e-16)).item(),
            "conf_obj": conf[obj].mean().item(), "conf_noobj": conf[noobj].mean().item(), "grid_size": G,
        })
        return out, total
