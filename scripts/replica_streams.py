"""Timing experiment: R sub-batches of 64/R tiles, each on its own stream and model instance, convolutions limited to
256/R workgroups (AY_CUS).  usage: AY_CUS=64 python scripts/replica_streams.py 4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from amyloid_yolo_paper_amd import cfg_gen, synth, parse_config
from amyloid_yolo_paper_amd.models import Darknet

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64 // R      # tiles per replica (default: a batch of 64 split R ways)
PREC = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda:0")
cfg = cfg_gen.write_cfg(3, "/tmp/cfg_rs")
models = []
params = synth.synth_params(parse_config.parse_model_config(cfg), seed=7)
for r in range(R):
    m = Darknet(cfg, img_size=1024, precision=PREC)
    sd = m.state_dict()
    for i, p in params.items():
        for k, name in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"),
                        ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
            if k in p:
                sd[f"module_list.{i}.{name}"].copy_(torch.from_numpy(p[k]))
    models.append(m.to(dev).eval())
xs = [torch.rand(B, 3, 1024, 1024, device=dev) for _ in range(R)]
streams = [torch.cuda.Stream() for _ in range(R)]

def step():
    for r in range(R):
        with torch.cuda.stream(streams[r]):
            models[r].forward_device(xs[r])

for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 6
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"R={R} B/replica={B} {PREC} AY_CUS={os.environ.get('AY_CUS')} ms per {R * B} tiles={dt*1e3:.2f} tiles/s={R * B/dt:.1f}")
