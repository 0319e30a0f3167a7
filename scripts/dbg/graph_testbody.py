"""The body of tests/test_gpu_configs.py::test_hip_graph_of_a_detection_step_replays_after_eager_steps with prints (AY_DYNAMIC=0)."""
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


def snapshot(res):
    return [t.clone() for t in res]


ref = [snapshot(step(x)) for x in xs]
torch.cuda.synchronize()
print("ref counts", [r[2].tolist() for r in ref], flush=True)
static_x = xs[0].clone()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    res = step(static_x)
torch.cuda.synchronize()
for k in (1, 2, 0, 1):
    j = 2 - k if k != 1 else 1
    eager = snapshot(step(xs[j]))
    if os.environ.get("SYNC_A") == "1":
        torch.cuda.synchronize()
    print("  eager on", j, "count", eager[2].tolist(), flush=True)
    del eager
    static_x.copy_(xs[k])
    if os.environ.get("SYNC_B") == "1":
        torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("replay on", k, "count", res[2].tolist(), "expected", ref[k][2].tolist(), flush=True)
