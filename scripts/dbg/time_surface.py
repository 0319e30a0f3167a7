import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from amyloid_yolo_paper_amd import cfg_gen, synth
from amyloid_yolo_paper_amd.models import Darknet
from amyloid_yolo_paper_amd.utils import non_max_suppression
dev = torch.device("cuda:0")
m = Darknet(cfg_gen.write_cfg(3, tempfile.mkdtemp()), precision="bf16").to(dev).eval()
B = 32
x = torch.from_numpy(synth.synth_tiles(4, 1024, start=0)).to(dev).repeat(B // 4, 1, 1, 1)
def T(f, n=3):
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print("   %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    return r
print("forward_device"); T(lambda: m.forward_device(x))
print("model(x) (device -> CPU tensor)"); out = T(lambda: m(x))
print("non_max_suppression(out)"); T(lambda: non_max_suppression(m(x), 0.5, 0.4))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); non_max_suppression(m(x), 0.5, 0.4); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(14)
