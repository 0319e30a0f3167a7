"""Darknet (YOLOv3) host class for the MI355X path.

Mirrors the reference's ``models.py`` call surface -- ``Darknet(config_path, img_size)``,
``model(x)`` / ``model(x, targets)``, ``load_darknet_weights`` / ``save_darknet_weights``,
``state_dict`` key names, ``.yolo_layers[i].metrics`` -- (reference ``models.py:225-336``) while the
forward itself is a planned sequence of calls into ``libamyloid_yolo_hip.so``:

* ``precision="bf16"`` (default): fp32 stem -> blocked-bf16 MFMA convolutions with fused
  BN-affine/LeakyReLU/shortcut epilogues, route+upsample gather, fp32 linear heads, fused decode.
* ``precision="fp16"``: the same plan and kernels on IEEE half storage (``v_mfma_*_f16``; inference only -- BASELINE.json
  configs[4]'s "fp16 MFMA path"): identical bytes and MFMA rate, an 8x smaller rounding step per stored activation.
* ``precision="fp32"``: every block through the fp32 nchw kernel (reference layout, reference
  precision) -- the mode the 1e-4 parity tests run.

The ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects in ``module_list`` are parameter containers only (they give
the reference's ``state_dict`` schema, optimiser hooks and ``.to(device)``); they are never called.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import ConvDesc, check, ptr
from .parse_config import parse_model_config


class Upsample(nn.Module):
    """nearest x``scale_factor`` placeholder (reference ``models.py:86-96``); executed inside the plan."""

    def __init__(self, scale_factor, mode="nearest"):
        super().__init__()
        self.scale_factor, self.mode = scale_factor, mode


class EmptyLayer(nn.Module):
    """route / shortcut placeholder."""


class YOLOLayer(nn.Module):
    """Detection-layer descriptor (anchors, class count, metrics dict); reference ``models.py:98-125``."""

    def __init__(self, anchors, num_classes, img_dim=416):
        super().__init__()
        self.anchors = anchors
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.ignore_thres = 0.5
        self.obj_scale = 1
        self.noobj_scale = 100
        self.metrics = {}
        self.img_dim = img_dim
        self.grid_size = 0


def create_modules(module_defs):
    """cfg blocks -> (hyperparams, ModuleList) with the reference's module names (``models.py:16-83``)."""
    hyperparams = module_defs.pop(0)
    filters_out = [int(hyperparams["channels"])]
    module_list = nn.ModuleList()
    for i, d in enumerate(module_defs):
        seq = nn.Sequential()
        kind = d["type"]
        if kind == "convolutional":
            bn = int(d["batch_normalize"])
            filters = int(d["filters"])
            k = int(d["size"])
            seq.add_module(f"conv_{i}", nn.Conv2d(filters_out[-1], filters, k, int(d["stride"]), (k - 1) // 2, bias=not bn))
            if bn:
                seq.add_module(f"batch_norm_{i}", nn.BatchNorm2d(filters, momentum=0.9, eps=1e-5))
            if d["activation"] == "leaky":
                seq.add_module(f"leaky_{i}", nn.LeakyReLU(0.1))
        elif kind == "maxpool":
            raise NotImplementedError("maxpool (yolov3-tiny) is outside the accelerated path (SURVEY.md §2)")
        elif kind == "upsample":
            filters = filters_out[-1]
            seq.add_module(f"upsample_{i}", Upsample(int(d["stride"])))
        elif kind == "route":
            filters = sum(filters_out[1:][int(j)] for j in d["layers"].split(","))
            seq.add_module(f"route_{i}", EmptyLayer())
        elif kind == "shortcut":
            filters = filters_out[1:][int(d["from"])]
            seq.add_module(f"shortcut_{i}", EmptyLayer())
        elif kind == "yolo":
            filters = filters_out[-1]
            mask = [int(v) for v in d["mask"].split(",")]
            flat = [int(v) for v in d["anchors"].split(",")]
            anchors = [(flat[2 * m], flat[2 * m + 1]) for m in mask]
            seq.add_module(f"yolo_{i}", YOLOLayer(anchors, int(d["classes"]), int(hyperparams["height"])))
        else:
            raise ValueError(f"unknown cfg block [{kind}]")
        module_list.append(seq)
        filters_out.append(filters)
    return hyperparams, module_list


def _pad_to(v, m):
    return (v + m - 1) // m * m


class _Plan:
    """owner of an ``ay_plan`` handle and its workspace"""

    def __init__(self, handle, workspace, ops, value_bytes):
        self.handle, self.workspace, self.ops, self.value_bytes = handle, workspace, ops, value_bytes

    def __del__(self):
        try:
            _lib.lib().ay_plan_destroy(self.handle)
        except Exception:
            pass


class Darknet(nn.Module):
    """YOLOv3 detector; see module docstring."""

    def __init__(self, config_path, img_size=416, precision="bf16"):
        super().__init__()
        self.module_defs = parse_model_config(config_path)
        self.hyperparams, self.module_list = create_modules(self.module_defs)
        self.yolo_layers = [m[0] for m in self.module_list if isinstance(m[0], YOLOLayer)]
        self.img_size = img_size
        self.seen = 0
        self.header_info = np.array([0, 0, 0, self.seen, 0], dtype=np.int32)
        assert precision in ("bf16", "fp16", "fp32")
        self.precision = precision
        # residual blocks run through the fused kernel: C=64 (measured 1.74 ms vs 2.37 for the two launches at B=64, 512^2);
        # the C=128 kernel exists (ay_resblock_supported) but measures 1.15 vs 1.04 ms, so it is not used by default
        self.fuse_block_channels = tuple(int(v) for v in os.environ.get("AY_FUSE_BLOCK_CHANNELS", "64").split(",") if v)
        self._graph = self._analyse()
        self._prep = None       # packed weights / folded BN, keyed by parameter versions
        self._act_bufs = {}      # (precision, B, S) -> per-layer device tensors
        self.keep_layer_outputs = False
        self.layer_outputs = None
        self.stem_mode = "fused_bf16"   # "fp32": separate fp32 stem kernel (layer 0 output materialised)
        self.fold_routes = True         # route [upsampled | direct] -> 1x1 conv without materialising the concatenation
        self.fuse_blocks = True         # fused residual-block kernel for the C=64/128 blocks (False: two conv launches)
        self.use_plan = os.environ.get("AY_USE_PLAN", "1") != "0"  # bf16 inference: the whole network issued by the native plan (ay_plan_forward)
        self.box_loss = "mse"           # "giou": 1 - GIoU replaces the four squared-error box terms (new feature; the
                                        # reference has only the MSE form, models.py:183-186)

    # ------------------------------------------------------------------ graph analysis
    def _analyse(self):
        """Static per-layer info: channels, resolved source indices, fusion decisions."""
        defs = self.module_defs
        n = len(defs)
        ch = []
        info = []
        cin = int(self.hyperparams["channels"])
        scale = []  # log2 downsample factor relative to the input
        for i, d in enumerate(defs):
            t = d["type"]
            e = {"type": t}
            prev_ch = ch[-1] if ch else cin
            prev_sc = scale[-1] if scale else 0
            if t == "convolutional":
                e.update(cin=prev_ch, cout=int(d["filters"]), k=int(d["size"]), stride=int(d["stride"]),
                         bn=bool(int(d["batch_normalize"])), leaky=d["activation"] == "leaky", src=i - 1)
                c, s = e["cout"], prev_sc + (1 if e["stride"] == 2 else 0)
            elif t == "upsample":
                assert int(d["stride"]) == 2, "only x2 nearest upsample is supported"
                e.update(src=i - 1)
                c, s = prev_ch, prev_sc - 1
            elif t == "route":
                srcs = [int(j) for j in d["layers"].split(",")]
                srcs = [j if j >= 0 else i + j for j in srcs]
                e.update(srcs=srcs)
                c, s = sum(ch[j] for j in srcs), scale[srcs[0]]
                assert all(scale[j] == s for j in srcs)
            elif t == "shortcut":
                j = int(d["from"])
                e.update(a=i - 1, b=j if j >= 0 else i + j)
                c, s = prev_ch, prev_sc
            elif t == "yolo":
                e.update(src=i - 1)
                c, s = prev_ch, prev_sc
            ch.append(c)
            scale.append(s)
            e.update(channels=c, log2_down=s)
            info.append(e)
        # consumers of every layer output
        users = {i: [] for i in range(n)}
        for i, e in enumerate(info):
            for j in ([e["src"]] if "src" in e else []) + e.get("srcs", []) + ([e["a"], e["b"]] if "a" in e else []):
                if j >= 0:
                    users[j].append(i)
        for i, e in enumerate(info):
            # conv whose only consumer is the next shortcut (as its "-1" operand): fuse the add into the epilogue
            e["fuse_into_shortcut"] = (e["type"] == "convolutional" and users[i] == [i + 1] and i + 1 < n
                                       and info[i + 1]["type"] == "shortcut" and info[i + 1]["a"] == i
                                       and info[i + 1]["b"] != i)
        # residual block conv1x1 (C -> C/2) -> conv3x3 (C/2 -> C) + shortcut from the 1x1's input: one fused kernel where
        # one exists (ay_resblock_fwd_bf16: C = 64, 128 -- the HBM-bound early blocks); the 1x1's output stays in LDS
        for i, e in enumerate(info):
            e["fuse_block"] = False
        for i, e in enumerate(info):
            if (e["type"] == "convolutional" and e["k"] == 1 and e["stride"] == 1 and e["bn"] and i + 2 < n
                    and users[i] == [i + 1] and info[i + 1]["type"] == "convolutional" and info[i + 1]["fuse_into_shortcut"]
                    and info[i + 1]["k"] == 3 and info[i + 1]["stride"] == 1 and info[i + 1]["bn"]
                    and info[i + 2]["b"] == e["src"] and e["src"] >= 0
                    and info[i + 1]["cout"] == e["cin"] == 2 * e["cout"] and e["cin"] in self.fuse_block_channels):
                e["fuse_block"] = True
                info[i + 1]["in_fused_block"] = True
        # layer 0 (3->32 3x3 s1) + layer 1 (32->64 3x3 s2): fused stem kernel, layer 0's output never materialised
        self._fuse_stem = (n > 1 and info[0]["type"] == "convolutional" and info[1]["type"] == "convolutional"
                           and (info[0]["cin"], info[0]["cout"], info[0]["k"], info[0]["stride"], info[0]["bn"]) == (3, 32, 3, 1, True)
                           and (info[1]["cout"], info[1]["k"], info[1]["stride"], info[1]["bn"]) == (64, 3, 2, True)
                           and users[0] == [1] and not info[1]["fuse_into_shortcut"])
        self._users = users
        return info

    # ------------------------------------------------------------------ weights I/O (reference models.py:257-336)
    def load_darknet_weights(self, weights_path):
        """Darknet binary: int32[5] header, then per conv block BN(beta,gamma,mean,var)|bias, then W (OIHW)."""
        with open(weights_path, "rb") as fh:
            header = np.fromfile(fh, dtype=np.int32, count=5)
            weights = np.fromfile(fh, dtype=np.float32)
        self.header_info = header
        self.seen = header[3]
        cutoff = 75 if "darknet53.conv.74" in weights_path else None
        pos = 0

        def take(t):
            nonlocal pos
            n = t.numel()
            t.data.copy_(torch.from_numpy(weights[pos:pos + n]).view_as(t))
            pos += n

        for i, (d, m) in enumerate(zip(self.module_defs, self.module_list)):
            if i == cutoff:
                break
            if d["type"] != "convolutional":
                continue
            conv = m[0]
            if int(d["batch_normalize"]):
                bn = m[1]
                for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                    take(t)
            else:
                take(conv.bias)
            take(conv.weight)
        self._prep = None

    def save_darknet_weights(self, path, cutoff=-1):
        with open(path, "wb") as fh:
            self.header_info[3] = self.seen
            self.header_info.tofile(fh)
            for d, m in zip(self.module_defs[:cutoff], self.module_list[:cutoff]):
                if d["type"] != "convolutional":
                    continue
                conv = m[0]
                if int(d["batch_normalize"]):
                    bn = m[1]
                    for t in (bn.bias, bn.weight, bn.running_mean, bn.running_var):
                        t.data.cpu().numpy().tofile(fh)
                else:
                    conv.bias.data.cpu().numpy().tofile(fh)
                conv.weight.data.cpu().numpy().tofile(fh)

    # ------------------------------------------------------------------ parameter preparation
    def _param_signature(self):
        sig = []
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t.data_ptr(), t._version))
        return hash(tuple(sig))

    def _prepare(self, device):
        sig = (self._param_signature(), self.precision, str(device))
        if self._prep is not None and self._prep["sig"] == sig:
            return self._prep
        L = _lib.lib()
        st = _lib.stream_ptr()
        prep = {"sig": sig, "layers": {}}
        for i, e in enumerate(self._graph):
            if e["type"] != "convolutional":
                continue
            m = self.module_list[i]
            conv = m[0]
            w = conv.weight.detach().to(device=device, dtype=torch.float32).contiguous()
            cout = e["cout"]
            first = (i == 0 and e["cin"] == 3 and e["k"] == 3 and e["stride"] == 1 and cout == 32)
            cpad = cout if self.precision == "fp32" else _pad_to(cout, 32)
            scale = torch.empty(cpad, device=device, dtype=torch.float32)
            shift = torch.empty(cpad, device=device, dtype=torch.float32)
            if e["bn"]:
                bn = m[1]
                g, b_, mu, var = (t.detach().to(device=device, dtype=torch.float32).contiguous()
                                  for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
                check(L.ay_fold_bn(ptr(g), ptr(b_), ptr(mu), ptr(var), None, C.c_float(bn.eps), ptr(scale), ptr(shift), cout, cpad, st),
                      "ay_fold_bn")
            else:
                bias = conv.bias.detach().to(device=device, dtype=torch.float32).contiguous()
                check(L.ay_fold_bn(None, None, None, None, ptr(bias), C.c_float(0.0), ptr(scale), ptr(shift), cout, cpad, st),
                      "ay_fold_bn")
            entry = dict(scale=scale, shift=shift, cpad=cpad, w=w, stem=first and self._mfma)
            if entry["stem"]:  # [32][32] 16-bit A-operand image of the stem filters for the fused stem kernel (k 27..31 = 0)
                w0 = torch.zeros(32, 32, device=device, dtype=torch.float32)
                w0[:, :27] = w.reshape(32, 27)
                entry["w0_bf16"] = w0.to(self._act_dtype).contiguous()
            if self._mfma and not entry["stem"]:
                assert e["cin"] % 16 == 0, f"layer {i}: cin {e['cin']} is not a multiple of 16"
                # blocked bf16 values are sized and strided in 16-channel planes of ceil16(channels); the kernels write
                # cout_pad = ceil32(cout) channels: a BN layer with filters % 32 == 16 would write one plane per image too many
                assert not e["bn"] or cout % 32 == 0, f"layer {i}: the bf16 path needs a multiple of 32 filters in conv+BN layers (got {cout}); use precision='fp32'"
                nbytes = L.ay_packed_weight_bytes(cpad, e["cin"], e["k"])
                packed = torch.empty(nbytes, device=device, dtype=torch.uint8)
                check(self._fn("ay_pack_conv_weights")(ptr(w), ptr(packed), cout, cpad, e["cin"], e["k"], st), "ay_pack_conv_weights")
                entry["packed"] = packed
            prep["layers"][i] = entry
        torch.cuda.current_stream().synchronize()  # temporaries (g,b_,mu,var,bias) die here
        self._prep = prep
        return prep

    # ------------------------------------------------------------------ storage type of the MFMA path
    @property
    def _mfma(self):
        return self.precision in ("bf16", "fp16")

    @property
    def _act_dtype(self):
        return torch.float16 if self.precision == "fp16" else torch.bfloat16

    def _fn(self, stem):
        """entry point ``stem``_bf16 | ``stem``_f16 of the library for this model's 16-bit storage type"""
        L = _lib.lib()
        if stem in ("ay_stem_s2_fused_fwd", "ay_stem_conv_fwd"):  # the bfloat16 forms carry no suffix
            return getattr(L, stem + ("_f16" if self.precision == "fp16" else ""))
        return getattr(L, stem + ("_f16" if self.precision == "fp16" else "_bf16"))

    def _unit(self, n, dev):
        """cached (ones[n], zeros[n]) device vectors: identity scale/shift for raw convolutions"""
        cache = self.__dict__.setdefault("_unit_cache", {})
        key = (n, str(dev))
        if key not in cache:
            cache[key] = (torch.ones(n, device=dev, dtype=torch.float32), torch.zeros(n, device=dev, dtype=torch.float32))
        return cache[key]

    # ------------------------------------------------------------------ forward
    def num_boxes(self, S):
        return sum(y.num_anchors * (S >> e["log2_down"]) ** 2 for y, e in
                   zip(self.yolo_layers, [g for g in self._graph if g["type"] == "yolo"]))

    def forward(self, x, targets=None):
        """Reference semantics (``models.py:237-255``): returns the CPU tensor ``[B, N, 5+C]``; the device copy is
        kept as ``out._ay_device`` so ``non_max_suppression`` does not upload it again."""
        if targets is not None:
            # training step: loss carries the autograd node whose backward runs the HIP dgrad / wgrad / BN / loss kernels
            # and fills every parameter's .grad.  precision="bf16": matrix-core path (train_engine_bf16.py);
            # precision="fp32": reference-precision path pinned by the golden training fixtures (train_engine.py).
            self._no_fp16_training()
            if self.precision == "bf16" and self.training:
                from .train_engine_bf16 import TrainStepBf16 as Step
            else:
                from .train_engine import TrainStep as Step
            loss, dev_out = Step.apply(self, x, targets, *self.parameters())
            return loss, dev_out.detach().cpu()
        if self.training:
            self._no_fp16_training()
            if self.precision == "bf16":
                from .train_engine_bf16 import train_forward_bf16 as fwd
            else:
                from .train_engine import train_forward as fwd
            with torch.no_grad():
                _, dev_out, _ = fwd(self, x, None)
            return dev_out.detach().cpu()
        dev_out = self.forward_device(x)
        out = dev_out.detach().cpu()
        self._gen = getattr(self, "_gen", 0) + 1
        dev_out._ay_gen = out._ay_gen = self._gen  # lets NMS tell that the device buffer still holds THIS output
        out._ay_device = dev_out
        return out

    def _no_fp16_training(self):
        if self.precision == "fp16":
            raise _lib.AyError("precision='fp16' is an inference storage type (no loss scaling, 5-bit exponent): train with "
                               "precision='bf16' (or 'fp32') and switch to fp16 for detection -- the weights are fp32 either way")

    def train_step_device(self, x, targets):
        """training forward with the autograd node attached, outputs left on the device: (loss, out [B,N,5+C] cuda)"""
        self._no_fp16_training()
        if self.precision == "bf16" and self.training:
            from .train_engine_bf16 import TrainStepBf16 as Step
        else:
            from .train_engine import TrainStep as Step
        return Step.apply(self, x, targets, *self.parameters())

    @torch.no_grad()
    def forward_device(self, x, out_slot=0):
        """x [B,3,S,S] float32 (any device) -> device tensor [B, N, 5+C] (valid until the next forward into the same
        ``out_slot``; two slots let NMS of batch i run on a side stream while batch i+1 computes)."""
        if not torch.cuda.is_available():
            raise _lib.AyError("no HIP device: the amyloid-yolo hot path has no CPU fallback")
        L = _lib.lib()
        dev = torch.device("cuda", torch.cuda.current_device())
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        B, Cin, S, S2 = x.shape
        assert S == S2 and S % 32 == 0, "square inputs with side divisible by 32 (reference models.py:238, datasets.py:78)"
        assert Cin == int(self.hyperparams["channels"])
        prep = self._prepare(dev)
        st = _lib.stream_ptr()
        bf16 = self._mfma   # the 16-bit MFMA path, bfloat16 or half storage
        key = (self.precision, B, S)
        bufs = self._act_bufs.setdefault(key, {})
        C_ = self.yolo_layers[0].num_classes
        N = self.num_boxes(S)
        okey = ("out", out_slot)
        if okey not in bufs:
            bufs[okey] = torch.empty(B, N, 5 + C_, device=dev, dtype=torch.float32)
        out = bufs[okey]

        def size_of(i):
            return S >> self._graph[i]["log2_down"] if i >= 0 else S

        def buf(i, f32=False, ch=None):
            """persistent output buffer of layer i"""
            if i not in bufs:
                e = self._graph[i]
                h = size_of(i)
                c = e["channels"] if ch is None else ch
                if not bf16:
                    bufs[i] = torch.empty(B, c, h, h, device=dev, dtype=torch.float32)
                elif f32:
                    bufs[i] = torch.empty(B, _pad_to(c, 32) // 16, h, h, 16, device=dev, dtype=torch.float32)
                else:
                    bufs[i] = torch.empty(B, _pad_to(c, 16) // 16, h, h, 16, device=dev, dtype=self._act_dtype)
            return bufs[i]

        prof = getattr(self, "profile_layers", None)  # bench.py: bracket these conv launches with HIP events
        if bf16 and self.use_plan and not self.keep_layer_outputs and prof is None:
            # the whole network from native code: one call, activations in one arena with lifetime reuse
            plan = self._plan(B, S, prep, dev)
            check(L.ay_plan_forward(plan.handle, ptr(x), ptr(plan.workspace), ptr(out), st), "ay_plan_forward")
            return out

        # values: ("t", tensor) materialised, ("up", layer) lazily upsampled view of another layer
        val = {}

        def resolve(i):
            """materialised tensor of layer i's output"""
            v = val[i]
            if v[0] == "t":
                return v[1]
            src = resolve(v[1])  # lazy upsample -> materialise
            c = self._graph[i]["channels"]
            o = buf(i)
            h = size_of(i)
            if bf16:
                check(L.ay_concat_upsample_bf16(ptr(src), c, 1, None, 0, ptr(o), B, h, h, st), "ay_concat_upsample_bf16")
            else:
                o.copy_(src.repeat_interleave(2, 2).repeat_interleave(2, 3))
            val[i] = ("t", o)
            return o

        row = 0
        for i, e in enumerate(self._graph):
            t = e["type"]
            if t == "convolutional":
                p = prep["layers"][i]
                hin, hout = size_of(e["src"]), size_of(i)
                fuse = e["fuse_into_shortcut"]
                res = resolve(self._graph[i + 1]["b"]) if fuse else None
                is_head = not e["bn"] and not e["leaky"]
                d = ConvDesc(B, e["cin"], e["cout"], hin, hin, hout, hout, e["k"], e["stride"], int(e["leaky"]),
                             int(bf16 and is_head), p["cpad"])
                tgt = i + 1 if fuse else i
                if i in val and val[i][0] == "fused":
                    continue  # second half of a fused residual block
                if bf16 and e["fuse_block"] and self.fuse_blocks:
                    p2 = prep["layers"][i + 1]
                    e2 = self._graph[i + 1]
                    assert L.ay_resblock_supported(e["cin"])
                    xin = resolve(e["src"])
                    o = buf(i + 2)
                    check(self._fn("ay_resblock_fwd")(ptr(xin), ptr(p["packed"]), ptr(p["scale"]), ptr(p["shift"]), int(e["leaky"]),
                                                 ptr(p2["packed"]), ptr(p2["scale"]), ptr(p2["shift"]), int(e2["leaky"]), ptr(o), B,
                                                 e["cin"], hout, hout, st), "ay_resblock_fwd_bf16")
                    val[i] = ("fused", None)
                    val[i + 1] = ("fused", None)
                    val[i + 2] = ("t", o)
                    continue
                if bf16 and i == 0 and self._fuse_stem and self.stem_mode == "fused_bf16":
                    val[i] = ("fused", None)
                    continue
                if bf16 and i == 1 and val.get(0, (None,))[0] == "fused":
                    p0 = prep["layers"][0]
                    o = buf(tgt)
                    check(self._fn("ay_stem_s2_fused_fwd")(ptr(x), ptr(p0["w0_bf16"]), ptr(p0["scale"]), ptr(p0["shift"]), int(self._graph[0]["leaky"]),
                                                 ptr(p["packed"]), ptr(p["scale"]), ptr(p["shift"]), int(e["leaky"]), ptr(o), B, S, S, st),
                          "ay_stem_s2_fused_fwd")
                    val[i] = ("t", o)
                    continue
                if bf16:
                    if p["stem"]:
                        o = buf(tgt)
                        check(self._fn("ay_stem_conv_fwd")(ptr(x), ptr(p["w"]), ptr(p["scale"]), ptr(p["shift"]), ptr(o), B, S, S,
                                                 int(e["leaky"]), st), "ay_stem_conv_fwd")
                    elif e["src"] >= 0 and val[e["src"]][0] == "catup":
                        ra, rb = val[e["src"]][1]
                        s1 = resolve(val[ra][1])      # half-resolution source of the lazy upsample
                        s2 = resolve(rb)
                        o = buf(tgt)
                        check(self._fn("ay_conv1x1_cat_fwd")(C.byref(d), ptr(s1), self._graph[ra]["channels"], ptr(s2), ptr(p["packed"]),
                                                        ptr(p["scale"]), ptr(p["shift"]), ptr(o), st), "ay_conv1x1_cat_fwd_bf16")
                    else:
                        src = x if e["src"] < 0 else resolve(e["src"])
                        if e["src"] < 0:
                            raise NotImplementedError("bf16 path expects the 3->32 3x3 stem as layer 0")
                        o = buf(tgt, f32=is_head)
                        timed = prof is not None and i in prof
                        if timed:
                            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            ev0.record()
                        check(self._fn("ay_conv_fwd")(C.byref(d), ptr(src), ptr(p["packed"]), ptr(p["scale"]), ptr(p["shift"]),
                                                 ptr(res), ptr(o), st), "ay_conv_fwd_bf16")
                        if timed:
                            ev1.record()
                            self.profile_events.append((i, ev0, ev1))
                else:
                    # route/upsample folded into the loader when the source is a lazy value
                    s1, c1, up1, s2 = self._f32_sources(e["src"], x, val, resolve)
                    o = buf(tgt)
                    check(L.ay_conv_fwd_f32(C.byref(d), ptr(s1), c1, up1, ptr(s2), ptr(p["w"]), ptr(p["scale"]), ptr(p["shift"]),
                                            ptr(res), ptr(o), st), "ay_conv_fwd_f32")
                val[i] = ("t", o)
                if fuse:
                    val[i] = ("fused", None)
                    val[i + 1] = ("t", o)
            elif t == "shortcut":
                if i in val:
                    continue  # produced by the fused epilogue of the previous conv
                raise NotImplementedError(f"layer {i}: unfused shortcut (a source other than the preceding conv)")
            elif t == "upsample":
                val[i] = ("up", e["src"])
            elif t == "route":
                srcs = e["srcs"]
                if len(srcs) == 1:
                    val[i] = val[srcs[0]] if val[srcs[0]][0] != "up" else ("t", resolve(srcs[0]))
                elif len(srcs) == 2:
                    if bf16 and self._cat_foldable(i, val):
                        val[i] = ("catup", srcs)  # [upsampled x2 | direct]: folded into the loader of the next 1x1 conv
                    elif bf16:
                        a, b_ = srcs
                        up = val[a][0] == "up"
                        s1 = resolve(val[a][1]) if up else resolve(a)
                        s2 = resolve(b_)
                        o = buf(i)
                        h = size_of(i)
                        check(L.ay_concat_upsample_bf16(ptr(s1), self._graph[a]["channels"], int(up), ptr(s2),
                                                        self._graph[b_]["channels"], ptr(o), B, h, h, st), "ay_concat_upsample_bf16")
                        val[i] = ("t", o)
                    else:
                        val[i] = ("cat", srcs)  # consumed by the next conv's loader
                else:
                    raise NotImplementedError("route with more than two sources")
            elif t == "yolo":
                y = self.module_list[i][0]
                head = resolve(e["src"])
                G = size_of(i)
                anchors = (C.c_float * (2 * y.num_anchors))(*[float(v) for a in y.anchors for v in a])
                check(L.ay_yolo_decode(ptr(head), 1 if bf16 else 0, ptr(out), B, y.num_anchors, y.num_classes, G, S, anchors, N,
                                       row, st), "ay_yolo_decode")
                y.grid_size, y.img_dim = G, S
                row += y.num_anchors * G * G
                val[i] = ("t", head)
        if self.keep_layer_outputs:
            self.layer_outputs = {i: v[1] for i, v in val.items() if v[0] == "t"}
        return out

    # ------------------------------------------------------------------ native plan
    def _lower(self, B, S, prep):
        """The bf16 walk of ``forward_device`` with symbolic values: the op list (``_lib.PlanOp``) and the byte size of every
        value, for ``ay_plan_create``.  Same decisions as the per-layer walk (fused stem, fused block, route folded into the 1x1
        loader, lazy upsample), so both issue the same kernels with the same arguments."""
        assert self._mfma
        ops, vbytes, val = [], [], {}
        N = self.num_boxes(S)

        def size_of(i):
            return S >> self._graph[i]["log2_down"] if i >= 0 else S

        def new_value(i, f32=False):
            c, h = self._graph[i]["channels"], size_of(i)
            vbytes.append(B * (_pad_to(c, 32) // 16) * h * h * 16 * 4 if f32 else B * (_pad_to(c, 16) // 16) * h * h * 16 * 2)
            return len(vbytes) - 1

        def op(kind, **kw):
            o = _lib.PlanOp()
            o.kind = kind
            o.src = o.src2 = o.res = o.dst = _lib.PLAN_NONE
            for k, v in kw.items():
                setattr(o, k, v)
            ops.append(o)
            return o

        def params(o, p, second=None):
            o.w, o.scale, o.shift = p["packed"].data_ptr() if "packed" in p else p["w"].data_ptr(), p["scale"].data_ptr(), p["shift"].data_ptr()
            if second is not None:
                o.w2, o.scale2, o.shift2 = second["packed"].data_ptr(), second["scale"].data_ptr(), second["shift"].data_ptr()

        def resolve(i):
            v = val[i]
            if v[0] == "t":
                return v[1]
            src = resolve(v[1])  # lazy upsample -> materialise
            h = size_of(i)
            dst = new_value(i)
            op(_lib.OP_CONCAT_UPSAMPLE, src=src, dst=dst, c1=self._graph[i]["channels"], up1=1, c2=0,
               conv=ConvDesc(B, 0, 0, h, h, h, h, 1, 1, 0, 0, 0))
            val[i] = ("t", dst)
            return dst

        row = 0
        for i, e in enumerate(self._graph):
            t = e["type"]
            if t == "convolutional":
                if i in val and val[i][0] == "fused":
                    continue  # second half of a fused residual block
                p = prep["layers"][i]
                hin, hout = size_of(e["src"]), size_of(i)
                fuse = e["fuse_into_shortcut"]
                is_head = not e["bn"] and not e["leaky"]
                d = ConvDesc(B, e["cin"], e["cout"], hin, hin, hout, hout, e["k"], e["stride"], int(e["leaky"]), int(is_head), p["cpad"])
                tgt = i + 1 if fuse else i
                if e["fuse_block"] and self.fuse_blocks:
                    p2, e2 = prep["layers"][i + 1], self._graph[i + 1]
                    src = resolve(e["src"])
                    dst = new_value(i + 2)
                    o = op(_lib.OP_RESBLOCK, src=src, dst=dst, conv=d, leaky2=int(e2["leaky"]))
                    params(o, p, p2)
                    val[i] = val[i + 1] = ("fused", None)
                    val[i + 2] = ("t", dst)
                    continue
                if i == 0 and self._fuse_stem and self.stem_mode == "fused_bf16":
                    val[i] = ("fused", None)
                    continue
                if i == 1 and val.get(0, (None,))[0] == "fused":
                    p0 = prep["layers"][0]
                    dst = new_value(tgt)
                    o = op(_lib.OP_STEM_S2_FUSED, dst=dst, leaky2=int(e["leaky"]),
                           conv=ConvDesc(B, 3, 32, S, S, S, S, 3, 1, int(self._graph[0]["leaky"]), 0, 32))
                    o.w, o.scale, o.shift = p0["w0_bf16"].data_ptr(), p0["scale"].data_ptr(), p0["shift"].data_ptr()
                    o.w2, o.scale2, o.shift2 = p["packed"].data_ptr(), p["scale"].data_ptr(), p["shift"].data_ptr()
                    val[i] = ("t", dst)
                    continue
                if p["stem"]:
                    dst = new_value(tgt)
                    o = op(_lib.OP_STEM, dst=dst, conv=d)
                    params(o, p)
                elif e["src"] >= 0 and val[e["src"]][0] == "catup":
                    ra, rb = val[e["src"]][1]
                    s1 = resolve(val[ra][1])  # half-resolution source of the lazy upsample
                    s2 = resolve(rb)
                    dst = new_value(tgt)
                    o = op(_lib.OP_CONV1X1_CAT, src=s1, src2=s2, dst=dst, conv=d, c1=self._graph[ra]["channels"])
                    params(o, p)
                else:
                    if e["src"] < 0:
                        raise NotImplementedError("bf16 path expects the 3->32 3x3 stem as layer 0")
                    src = resolve(e["src"])
                    res = resolve(self._graph[i + 1]["b"]) if fuse else _lib.PLAN_NONE
                    dst = new_value(tgt, f32=is_head)
                    o = op(_lib.OP_CONV, src=src, res=res, dst=dst, conv=d)
                    params(o, p)
                    o._layer = i
                val[i] = ("t", dst)
                if fuse:
                    val[i] = ("fused", None)
                    val[i + 1] = ("t", dst)
            elif t == "shortcut":
                if i not in val:
                    raise NotImplementedError(f"layer {i}: unfused shortcut (a source other than the preceding conv)")
            elif t == "upsample":
                val[i] = ("up", e["src"])
            elif t == "route":
                srcs = e["srcs"]
                if len(srcs) == 1:
                    val[i] = val[srcs[0]] if val[srcs[0]][0] != "up" else ("t", resolve(srcs[0]))
                elif len(srcs) == 2:
                    if self._cat_foldable(i, val):
                        val[i] = ("catup", srcs)
                    else:
                        a, b_ = srcs
                        up = val[a][0] == "up"
                        s1 = resolve(val[a][1]) if up else resolve(a)
                        s2 = resolve(b_)
                        h = size_of(i)
                        dst = new_value(i)
                        op(_lib.OP_CONCAT_UPSAMPLE, src=s1, src2=s2, dst=dst, c1=self._graph[a]["channels"], up1=int(up),
                           c2=self._graph[b_]["channels"], conv=ConvDesc(B, 0, 0, h, h, h, h, 1, 1, 0, 0, 0))
                        val[i] = ("t", dst)
                else:
                    raise NotImplementedError("route with more than two sources")
            elif t == "yolo":
                y = self.module_list[i][0]
                head = resolve(e["src"])
                G = size_of(i)
                assert y.num_anchors <= 6
                o = op(_lib.OP_DECODE, src=head, num_anchors=y.num_anchors, num_classes=y.num_classes, grid=G, row_offset=row,
                       conv=ConvDesc(B, 0, 0, G, G, G, G, 1, 1, 0, 0, 0))
                for k, v in enumerate(float(v) for a in y.anchors for v in a):
                    o.anchors_wh[k] = v
                y.grid_size, y.img_dim = G, S
                row += y.num_anchors * G * G
                val[i] = ("t", head)
        assert row == N
        return ops, vbytes

    def _plan(self, B, S, prep, dev):
        """(plan handle, workspace tensor, op list) for this batch shape; plans live in ``prep`` and die with it (new weights)."""
        plans = prep.setdefault("plans", {})
        key = (B, S, self.fuse_blocks, self.fold_routes, self.stem_mode, tuple(self.fuse_block_channels))
        if key not in plans:
            L = _lib.lib()
            ops, vbytes = self._lower(B, S, prep)
            arr = (_lib.PlanOp * len(ops))(*ops)
            vb = (C.c_size_t * len(vbytes))(*vbytes)
            handle = C.c_void_p()
            check(L.ay_plan_create(arr, len(ops), vb, len(vbytes), S, self.num_boxes(S), int(self.precision == "fp16"), C.byref(handle)),
                  "ay_plan_create")
            ws = torch.empty(L.ay_plan_workspace_bytes(handle), device=dev, dtype=torch.uint8)
            plans[key] = _Plan(handle, ws, ops, sum(vbytes))
        return plans[key]

    def plan_profile_begin(self, B, S, layers=None, every=1):
        """start recording HIP event pairs around the ops of ``layers`` (conv layer indices; None = every op) in the plan of
        this batch shape, on every ``every``-th forward (bench.py: roofline of the 3x3 family)"""
        dev = torch.device("cuda", torch.cuda.current_device())
        plan = self._plan(B, S, self._prepare(dev), dev)
        sel = None
        if layers is not None:
            sel = (C.c_ubyte * len(plan.ops))(*[int(getattr(o, "_layer", None) in layers) for o in plan.ops])
        check(_lib.lib().ay_plan_profile_begin_every(plan.handle, sel, int(every)), "ay_plan_profile_begin")

    def plan_profile_end(self, B, S):
        """-> ([(layer index | None, op kind, ms summed over the recorded forwards)], number of forwards)"""
        dev = torch.device("cuda", torch.cuda.current_device())
        plan = self._plan(B, S, self._prepare(dev), dev)
        ms = (C.c_float * len(plan.ops))()
        n = C.c_int(0)
        check(_lib.lib().ay_plan_profile_end(plan.handle, ms, C.byref(n)), "ay_plan_profile_end")
        return [(getattr(o, "_layer", None), o.kind, float(ms[k])) for k, o in enumerate(plan.ops)], n.value

    def _cat_foldable(self, i, val):
        """route i = [lazily upsampled a | b] whose only consumer is the next layer, a 1x1 bf16 conv block the dual-source
        kernel covers (channel split in multiples of 64, padded cout a multiple of 128, not fused into a shortcut)"""
        if not self.fold_routes or self.keep_layer_outputs:
            return False
        a, b_ = self._graph[i]["srcs"]
        if val[a][0] != "up" or val[b_][0] != "t" or self._users[i] != [i + 1] or i + 1 >= len(self._graph):
            return False
        e = self._graph[i + 1]
        ca, cb = self._graph[a]["channels"], self._graph[b_]["channels"]
        return (e["type"] == "convolutional" and e["k"] == 1 and e["stride"] == 1 and e["bn"] and not e["fuse_into_shortcut"]
                and not e.get("fuse_block") and ca % 64 == 0 and cb % 64 == 0 and _pad_to(e["cout"], 32) % 128 == 0)

    def _f32_sources(self, src, x, val, resolve):
        """(src1, cin1, up1, src2) for the fp32 kernel, folding route/upsample chains."""
        if src < 0:
            return x, x.shape[1], 0, None
        v = val[src]
        if v[0] == "cat":
            a, b_ = v[1]
            up = val[a][0] == "up"
            s1 = resolve(val[a][1]) if up else resolve(a)
            return s1, self._graph[a]["channels"], int(up), resolve(b_)
        if v[0] == "up":
            s1 = resolve(v[1])
            return s1, self._graph[src]["channels"], 1, None
        t = resolve(src)
        return t, self._graph[src]["channels"], 0, None

    def layer_output_nchw(self, i):
        """fp32 NCHW copy of a kept layer output (tests); needs keep_layer_outputs=True before forward."""
        L = _lib.lib()
        t = self.layer_outputs[i]
        if t.dtype == torch.float32 and t.dim() == 4:
            return t.clone()
        B, P, H, W, _ = t.shape
        c = self._graph[i]["channels"]
        o = torch.empty(B, c, H, W, device=t.device, dtype=torch.float32)
        fn = {torch.bfloat16: L.ay_blocked_bf16_to_nchw_f32, torch.float16: L.ay_blocked_f16_to_nchw_f32}.get(t.dtype, L.ay_blocked_f32_to_nchw_f32)
        check(fn(ptr(t), ptr(o), B, c, H, W, _lib.stream_ptr()), "blocked_to_nchw")
        return o
