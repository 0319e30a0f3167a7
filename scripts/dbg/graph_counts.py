"""Replay a captured detection step on three inputs; print NMS counts of eager steps and replays (run with AY_DYNAMIC=0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
import golden_cases as gc
from test_gpu_parity import build_models
from amyloid_yolo_paper_amd.utils import nms_device

dev = torch.device("cuda", 0)
m, _ = build_models(3, "/tmp/graph_cfg", dev, "bf16")
S, B = 256, 2
xs = [torch.from_numpy(gc.model_inputs(S, B, start)).to(dev) for start in (0, 3, 5)]


def step(x):
    out = m.forward_device(x, out_slot=0)
    return nms_device(out, 0.5, 0.4, 512, slot=7)


for i, x in enumerate(xs):
    r = step(x)
    torch.cuda.synchronize()
    print("eager", i, "count", r[2].tolist(), "cand", r[3].tolist(), "out sum %.6f" % float(m.forward_device(x, out_slot=1).double().sum()), flush=True)
static_x = xs[0].clone()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    res = step(static_x)
torch.cuda.synchronize()
for k in (1, 2, 0, 1, 2):
    if os.environ.get("EAGER_BETWEEN", "1") == "1":
        step(xs[(k + 1) % 3])
    static_x.copy_(xs[k])
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    out0 = m._act_bufs[("bf16", B, S)][("out", 0)]
    print("replay", k, "count", res[2].tolist(), "cand", res[3].tolist(), "out(slot0) sum %.6f" % float(out0.double().sum()), flush=True)
