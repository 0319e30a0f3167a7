"""Stand-alone time of merge-NMS on synthetic predictions with more candidates per image than the LDS path holds (workspace path)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
import golden_cases as gc
from amyloid_yolo_paper_amd.utils import nms_device

dev = torch.device("cuda", 0)
for rows, n in ((64512, 900), (64512, 2000), (64512, 4000), (258048, 4000), (258048, 16000)):
    B = 8
    pred = torch.from_numpy(gc.nms_prediction(rows, [n] * B, 3, 7, size=2048.0)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for it in range(4):
        o = pred.clone()
        torch.cuda.synchronize()
        ev[0].record()
        r = nms_device(o, 0.5, 0.4, 8192, 1)
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print(f"rows {rows} candidates {n} x {B} images: heads/image {r[2].float().mean().item():.0f}, ms per call {min(ts):.3f}", flush=True)
