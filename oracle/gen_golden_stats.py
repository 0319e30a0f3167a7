"""Golden vectors for the evaluation statistics from the imported reference:
`get_batch_statistics` (utils/utils.py:154-190), `ap_per_class` / `compute_ap` (utils/utils.py:69-151).
Run in the build container:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden_stats"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(1, os.environ.get("AY_REFERENCE", "/root/reference"))
import golden_cases as gc  # noqa: E402
from utils import utils as ref_utils  # noqa: E402  (reference)

if __name__ == "__main__":
    out = {}
    for thr in (0.5, 0.75):
        outputs, targets = gc.stats_inputs()
        t_out = [None if o is None else torch.from_numpy(o) for o in outputs]
        metrics = ref_utils.get_batch_statistics(t_out, torch.from_numpy(targets), iou_threshold=thr)
        tag = f"t{int(thr * 100)}"
        out[f"{tag}_n"] = np.int64(len(metrics))
        for k, (tp, scores, labels) in enumerate(metrics):
            out[f"{tag}_tp{k}"] = np.asarray(tp, np.float64)
            out[f"{tag}_scores{k}"] = scores.numpy()
            out[f"{tag}_labels{k}"] = labels.numpy()
        tp, scores, labels = [np.concatenate([np.asarray(x) for x in col], 0) for col in zip(*metrics)]
        p, r, ap, f1, cls = ref_utils.ap_per_class(tp, scores, labels, targets[:, 1].tolist())
        out.update({f"{tag}_p": p, f"{tag}_r": r, f"{tag}_ap": ap, f"{tag}_f1": f1, f"{tag}_cls": cls})
        print(tag, "images with detections", len(metrics), "TP", int(tp.sum()), "AP", ap)
    path = os.path.join(REPO, "tests", "golden", "stats_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
