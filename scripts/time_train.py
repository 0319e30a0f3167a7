import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amyloid_yolo_paper_amd import cfg_gen, synth
from amyloid_yolo_paper_amd.models import Darknet
from amyloid_yolo_paper_amd.utils import weights_init_normal
B, S = int(sys.argv[1]), int(sys.argv[2])
m = Darknet(cfg_gen.write_cfg(3), precision=(sys.argv[3] if len(sys.argv) > 3 else 'fp32')).to('cuda'); m.apply(weights_init_normal); m.train()
x = torch.from_numpy(synth.synth_tiles(min(B, 4), S)).repeat((B + 3) // 4, 1, 1, 1)[:B].cuda()
tg = torch.from_numpy(synth.synth_targets(B, 3, seed=3, grid=S // 8)).cuda()
opt = torch.optim.Adam(m.parameters())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    loss, out = m(x, tg); torch.cuda.synchronize(); t1 = time.time()
    loss.backward(); torch.cuda.synchronize(); t2 = time.time()
    opt.step(); opt.zero_grad(); torch.cuda.synchronize(); t3 = time.time()
    print(f"B={B} S={S} it{it} fwd {t1-t0:.3f}s bwd {t2-t1:.3f}s opt {t3-t2:.3f}s loss {loss.item():.3f}  -> {B/(t3-t0):.2f} imgs/s", flush=True)
