"""Diagnosis: where does the error of the stem's weight gradient (bf16 training step vs teacher-forced oracle) come from?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, torch.nn.functional as F
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd.models import Darknet
from oracle.darknet_oracle import OracleDarknet
from test_gpu_train_bf16 import from_blocked, bf
C_, S, B = 3, 256, 4
d = "/tmp/cfgd"; os.makedirs(d, exist_ok=True)
cfg = cfg_gen.write_cfg(C_, d); defs = parse_config.parse_model_config(cfg)
wpath = os.path.join(d, "w.weights"); synth.write_darknet_weights(wpath, defs, synth.synth_params(defs, seed=7), seen=0)
x = torch.from_numpy(synth.synth_tiles(B, S, 10))
tg = torch.from_numpy(synth.synth_targets(B, C_, seed=21, max_per_tile=6, min_per_tile=3, wh_range=(0.05, 0.4), grid=S // 8))
m = Darknet(cfg, precision="bf16").to("cuda"); m.load_darknet_weights(wpath); m.train()
m._dbg_keep_dz = {0: None, 1: None, 2: None, 5: None}
loss, out = m(x, tg)
stt = loss.grad_fn.stt; graph = m._graph
forced = {"z": {}, "y": {}}
for i, rec in stt.conv.items():
    if rec["kind"] == "bn": forced["z"][i] = from_blocked(rec["z"], graph[i]["cout"])
for i, v in stt.val.items():
    if v is None or isinstance(v, tuple) or graph[i]["type"] not in ("convolutional", "shortcut"): continue
    forced["y"][i] = v.float().cpu() if v.dim() == 4 else from_blocked(v, graph[i]["channels"])
loss.backward()
o = OracleDarknet(cfg); o.load_darknet_weights(wpath); o.require_grad(); o.keep_z = {}
lo, _ = o.forward(x, tg, mode="bf16_train", train_bn=True, forced=forced); lo.backward()
for i in (0, 1, 2, 5):
    dz_h = from_blocked(m._dbg_keep_dz[i], graph[i]["cout"])
    dz_o = o.keep_z[i].grad
    print(f"layer {i}: dz relL2 {float((dz_h - dz_o).norm() / dz_o.norm()):.4f}  |dz| {float(dz_o.norm()):.3e}  sum/abs-sum per channel {float(dz_o.sum((0,2,3)).abs().mean() / dz_o.abs().sum((0,2,3)).mean()):.2e}")
dz_o, dz_h = o.keep_z[0].grad, from_blocked(m._dbg_keep_dz[0], 32)
xb = bf(x)
def wgrad(xx, dz):
    w = torch.zeros(32, 3, 3, 3, requires_grad=True)
    F.conv2d(xx, w, None, 1, 1).backward(dz)
    return w.grad
ref = o.params[0]["weight"].grad
got = m.module_list[0][0].weight.grad.float().cpu()
xc = x - x.mean((0, 2, 3), keepdim=True)
for name, g in (("HIP", got), ("cpu: bf16(x), oracle dz", wgrad(xb, dz_o)), ("cpu: bf16(x), HIP dz", wgrad(xb, dz_h)), ("cpu: bf16(x-c), HIP dz", wgrad(bf(xc), dz_h)),
                ("cpu: bf16(x-c), oracle dz", wgrad(bf(xc), dz_o)), ("cpu: x-c fp32, oracle dz", wgrad(xc, dz_o)), ("cpu: x fp32, oracle dz", wgrad(x, dz_o))):
    print(f"{name:28s} relL2 vs oracle dW {float((g - ref).norm() / ref.norm()):.4f}   |dW| {float(g.norm()):.4e}")
