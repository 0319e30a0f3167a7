"""Which part of a detection step breaks under HIP-graph replay?  usage: graph_bisect.py fwd|nms|all [size] [batch]
Builds the small bf16 model, captures the chosen part on a side stream, replays it 3 times with eager steps in between, compares."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
import golden_cases as gc
from test_gpu_parity import build_models
from amyloid_yolo_paper_amd.utils import nms_device

what = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
m, _ = build_models(3, "/tmp/graph_cfg", dev, "bf16")
if os.environ.get("AY_USE_PLAN_OFF"):
    m.use_plan = False
xs = [torch.from_numpy(gc.model_inputs(S, B, start)).to(dev) for start in (0, 3)]


def step(x, pre=None):
    if what == "fwd":
        return (m.forward_device(x, out_slot=0),)
    if what == "nms":
        return tuple(nms_device(pre, 0.5, 0.4, 512, slot=7))
    out = m.forward_device(x, out_slot=0)
    return tuple(nms_device(out, 0.5, 0.4, 512, slot=7))


pre = [m.forward_device(x, out_slot=1).clone() for x in xs]
ref = [[t.clone() for t in step(x, p.clone())] for x, p in zip(xs, pre)]
torch.cuda.synchronize()
static_x, static_pre = xs[0].clone(), pre[0].clone()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    res = step(static_x, static_pre)
torch.cuda.synchronize()
print("captured", what, flush=True)
for k in (1, 0, 1):
    step(xs[1 - k], pre[1 - k].clone())
    static_x.copy_(xs[k])
    static_pre.copy_(pre[k])
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(res[:1], ref[k][:1])) if what != "fwd" else torch.equal(res[0], ref[k][0])
    print("replay on input", k, "matches eager:", same, flush=True)
print("OK", what)
