"""Stand-alone time of merge-NMS on the bench's own decode output (B=64, 1024^2, ~490 candidates per tile): HIP events around
`nms_device` on an otherwise idle GPU (the bench overlaps it with the next batch's convolutions; this is the drop-in cost)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd.models import Darknet
from amyloid_yolo_paper_amd.utils import nms_device

dev = torch.device("cuda", 0)
B, S = int(os.environ.get("B", "64")), 1024
cfg = cfg_gen.write_cfg(3)
params = synth.synth_params(parse_config.parse_model_config(cfg), seed=7)
m = Darknet(cfg, img_size=S, precision="bf16")
sd = m.state_dict()
for i, p in params.items():
    for k, name in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"),
                    ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
        if k in p:
            sd[f"module_list.{i}.{name}"].copy_(torch.from_numpy(p[k]))
m = m.to(dev).eval()
x = torch.from_numpy(synth.synth_tiles(16, S)).to(dev).repeat(B // 16, 1, 1, 1).contiguous()
out = m.forward_device(x).clone()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for it in range(12):
    o = out.clone()
    torch.cuda.synchronize()
    ev[0].record()
    rows, keep, count, cand = nms_device(o, 0.5, 0.4, 2048, 0)
    ev[1].record()
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
print("nms_device B=%d: candidates/tile %.1f, heads/tile %.1f; ms per call (12 calls) min %.3f median %.3f" %
      (B, cand.float().mean().item(), count.float().mean().item(), min(ts), sorted(ts)[len(ts) // 2]))
