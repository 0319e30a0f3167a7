"""Layer-by-layer: stored activations of the bf16 training forward (train_engine_bf16.train_forward_bf16) against
OracleDarknet.forward(mode="bf16_train", train_bn=True) on the same batch.  Diagnosis tool (GPU)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import torch

from amyloid_yolo_paper_amd import _lib, cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd._lib import check, ptr
from amyloid_yolo_paper_amd.models import Darknet
from amyloid_yolo_paper_amd.train_engine_bf16 import train_forward_bf16
from oracle.darknet_oracle import OracleDarknet

C_, S, B = 3, int(os.environ.get("S", "256")), int(os.environ.get("B", "4"))
d = "/tmp/cfgd"
os.makedirs(d, exist_ok=True)
cfg = cfg_gen.write_cfg(C_, d)
defs = parse_config.parse_model_config(cfg)
wpath = os.path.join(d, "w.weights")
synth.write_darknet_weights(wpath, defs, synth.synth_params(defs, seed=7), seen=0)
x = torch.from_numpy(synth.synth_tiles(B, S, 10))
tg = torch.from_numpy(synth.synth_targets(B, C_, seed=21, max_per_tile=6, min_per_tile=3, wh_range=(0.05, 0.4), grid=S // 8))

o = OracleDarknet(cfg)
o.load_darknet_weights(wpath)
with torch.no_grad():
    lo, _ = o.forward(x, tg, mode="bf16_train", train_bn=True, collect=True)
ref = o.layer_outputs

m = Darknet(cfg, precision="bf16").to("cuda")
m.load_darknet_weights(wpath)
m.train()
with torch.no_grad():
    loss, out, stt = train_forward_bf16(m, x, tg)
print("loss HIP", float(loss), "oracle", float(lo))
L = _lib.lib()


def nchw(t, c):
    if t.dim() == 4:
        return t.float().cpu()
    Bb, P, H, W, _ = t.shape
    o_ = torch.empty(Bb, c, H, W, device=t.device, dtype=torch.float32)
    fn = L.ay_blocked_bf16_to_nchw_f32 if t.dtype == torch.bfloat16 else L.ay_blocked_f32_to_nchw_f32
    check(fn(ptr(t), ptr(o_), Bb, c, H, W, _lib.stream_ptr()))
    return o_.cpu()


for i, e in enumerate(m._graph):
    v = stt.val.get(i)
    if v is None or isinstance(v, tuple) or e["type"] == "yolo":
        continue
    got = nchw(v, e["channels"])
    r = ref[i]
    if got.shape != r.shape:
        continue
    diff = (got - r).abs()
    ulp = torch.clamp(r.abs(), min=1e-30) * 2.0 ** -8
    frac_gt1 = float((diff > ulp + 1e-6).float().mean())
    print(f"layer {i:3d} {e['type']:13s} max|d| {float(diff.max()):.3e}  rel L2 {float(diff.norm() / (r.norm() + 1e-30)):.3e}  frac>1ulp {frac_gt1:.2e}")
