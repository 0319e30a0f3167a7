import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, time
import golden_cases as gc
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd.models import Darknet
C_, S, B = 3, int(sys.argv[1]), int(sys.argv[2])
cfg = cfg_gen.write_cfg(C_, '/tmp/cfgd'); defs = parse_config.parse_model_config(cfg)
wpath = '/tmp/cfgd/w.weights'
synth.write_darknet_weights(wpath, defs, synth.synth_params(defs, seed=7), seen=0)
x = torch.from_numpy(synth.synth_tiles(B, S, 10)); tg = torch.from_numpy(synth.synth_targets(B, C_, seed=21, max_per_tile=6, min_per_tile=3, wh_range=(0.05, 0.4), grid=S // 8))
res = {}
for prec in ('fp32', 'bf16'):
    m = Darknet(cfg, precision=prec).to('cuda'); m.load_darknet_weights(wpath); m.train()
    torch.cuda.synchronize(); t0 = time.time()
    loss, out = m(x, tg); loss.backward(); torch.cuda.synchronize()
    print(prec, 'loss', loss.item(), 'time', time.time() - t0)
    res[prec] = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
for n in res['fp32']:
    if 'conv' in n and n.endswith('weight'):
        a, b = res['bf16'][n].reshape(-1), res['fp32'][n].reshape(-1)
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)); rel = float((a - b).norm() / (b.norm() + 1e-30))
        print(f'{n:40s} cos {cos:.4f} relL2 {rel:.3f} |g| {float(b.norm()):.3e}')
