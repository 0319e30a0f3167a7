"""Throughput of the WSI -> tile -> detection stream (§8f N4): a synthetic slide of TY x TX 1536-px tiles in host memory,
streamed strip by strip (pinned upload on a copy stream, device-side tiling + /255 + resize to 1024), model + merge-NMS.
usage: python scripts/bench_wsi.py [TY TX] ; prints tiles/s including the PCIe upload (this is NOT bench.py's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from amyloid_yolo_paper_amd import cfg_gen, synth
from amyloid_yolo_paper_amd.models import Darknet
from amyloid_yolo_paper_amd.wsi import RegionTileStream, detect_region
import tempfile

TY, TX = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 32)
tile, S = 1536, 1024
dev = torch.device("cuda:0")
from amyloid_yolo_paper_amd import parse_config
cfg = cfg_gen.write_cfg(3, tempfile.mkdtemp())
m = Darknet(cfg, precision="bf16")
sd = m.state_dict()   # the calibrated synthetic weights of bench.py (random-init nets put every box at conf ~0.5: NMS-bound nonsense)
for i, p in synth.synth_params(parse_config.parse_model_config(cfg), seed=7).items():
    for k, name in (("weight", f"conv_{i}.weight"), ("bias", f"conv_{i}.bias"), ("gamma", f"batch_norm_{i}.weight"),
                    ("beta", f"batch_norm_{i}.bias"), ("mean", f"batch_norm_{i}.running_mean"), ("var", f"batch_norm_{i}.running_var")):
        if k in p:
            sd[f"module_list.{i}.{name}"].copy_(torch.from_numpy(p[k]))
m = m.to(dev).eval()
base = synth.synth_tiles(4, 1536, start=0)                       # [4,3,1536,1536] float
base = (base * 255).astype(np.uint8).transpose(0, 2, 3, 1)
row = np.concatenate([base[i % 4] for i in range(TX)], 1)
raster = np.concatenate([np.roll(row, 97 * j, 1) for j in range(TY)], 0)
print("raster", raster.shape, "%.2f GB" % (raster.nbytes / 1e9), flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for tiles, cs in RegionTileStream(raster, tile, S):
        n += tiles.shape[0]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("ingest only: %.0f tiles/s (%.1f GB/s of slide)" % (n / (t1 - t0), raster.nbytes / (t1 - t0) / 1e9), flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = detect_region(m, raster, tile, S, conf_thres=0.5, nms_thres=0.4, batch_size=TX)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("detect_region: %.0f tiles/s (%d tiles, %d with detections)" % (TY * TX / (t1 - t0), TY * TX, len(res)), flush=True)
