"""Forward passes of the bench configuration WITHOUT merge-NMS (debug: how much the overlapped NMS costs the main stream)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from amyloid_yolo_paper_amd import cfg_gen, parse_config, synth
from amyloid_yolo_paper_amd.models import Darknet

dev = torch.device("cuda", 0)
cfg = cfg_gen.write_cfg(3)
defs = parse_config.parse_model_config(cfg)
params = synth.synth_params(defs, seed=7)
model = Darknet(cfg, img_size=1024, precision="bf16")
sd = model.state_dict()
for i, p in params.items():
    for k, name in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"),
                    ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
        if k in p:
            sd[f"module_list.{i}.{name}"].copy_(torch.from_numpy(p[k]))
model = model.to(dev).eval()
x = torch.from_numpy(synth.synth_tiles(8, 1024, start=0)).to(dev).repeat(8, 1, 1, 1).contiguous()
for _ in range(3):
    model.forward_device(x, out_slot=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for i in range(N):
    model.forward_device(x, out_slot=i & 1)
torch.cuda.synchronize()
print("forward only: %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))

# the same forward captured as a HIP graph and replayed through utils.graph_replay (replay + ay_stream_fence)
from amyloid_yolo_paper_amd.utils import graph_replay
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    model.forward_device(x, out_slot=0)
torch.cuda.synchronize()
for _ in range(4):
    graph_replay(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    graph_replay(g)
torch.cuda.synchronize()
print("forward only, graph replay: %.3f ms/step" % ((time.perf_counter() - t0) / N * 1e3))
