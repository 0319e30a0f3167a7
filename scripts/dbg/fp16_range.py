"""Where does the half-precision inference path leave its range after a few training steps from random init?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
from amyloid_yolo_paper_amd import cfg_gen, synth
from amyloid_yolo_paper_amd.models import Darknet
from amyloid_yolo_paper_amd.parallel import FlatAdam, FlatGradReducer
from amyloid_yolo_paper_amd.utils import weights_init_normal
from test_gpu_stress import dense_targets
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B, Cc = 2, 3
dev = torch.device("cuda", 0)
cfg = cfg_gen.write_cfg(Cc, "/tmp/fp16dbg")
torch.manual_seed(4321)
model = Darknet(cfg, img_size=S, precision="bf16").to(dev)
model.apply(weights_init_normal)
model.box_loss = "giou"
model.train()
model.collect_metrics = False
red = FlatGradReducer(model.parameters(), n_buckets=4).attach(model)
opt = FlatAdam(red)
x = torch.from_numpy(synth.synth_tiles(B, S, start=7)).to(dev)
tg = torch.from_numpy(dense_targets(B, Cc, 500, 92)).to(dev)
for _ in range(3):
    red.begin(); loss, _ = model.train_step_device(x, tg); loss.backward(); red.all_reduce(average=False); opt.step(); red.zero()
    print("loss", float(loss.item()))
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
for prec in ("fp32", "bf16", "fp16"):
    det = Darknet(cfg, img_size=S, precision=prec).to(dev)
    det.load_state_dict(sd)
    det.eval()
    det.keep_layer_outputs = True
    out = det.forward_device(x)
    print(prec, "finite", bool(torch.isfinite(out).all()), "nonfinite rows", int((~torch.isfinite(out)).any(-1).sum()))
    worst = []
    for li, t in sorted(det.layer_outputs.items()):
        f = t.float()
        worst.append((float(f[torch.isfinite(f)].abs().max()) if torch.isfinite(f).any() else float("nan"), int((~torch.isfinite(f)).sum()), li))
    print(prec, "largest |activation| per layer (top 8):", sorted(worst, reverse=True)[:8])
    print(prec, "first layers with non-finite values:", [(li, n) for m_, n, li in sorted(worst, key=lambda w: w[2]) if n][:6])
    bad = ~torch.isfinite(out)
    if bad.any():
        print(prec, "non-finite columns:", bad.any(0).any(0).tolist())
    del det
