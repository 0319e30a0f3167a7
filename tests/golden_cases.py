"""Seeded input generators shared by oracle/gen_golden.py (which feeds them to the imported
reference to produce tests/golden/*.npz) and by the tests (which feed them to the oracle and to
the HIP path).  Only expected OUTPUTS are stored in the fixtures; inputs are regenerated here."""
import numpy as np

NMS_CASES = [
    # name, rows per image, candidates per image (list = batch), classes, seed
    ("empty", 256, [0, 0], 2, 1),
    ("single", 256, [1, 0, 1], 2, 2),
    ("seven", 512, [7, 3], 3, 3),
    ("wave64", 1024, [64, 65, 63], 2, 4),
    ("n500", 4096, [500, 431], 3, 5),
    ("n4096", 10647, [4096], 2, 16),
    ("dups_crossclass", 128, [24, 24], 3, 7),
    ("conf_low_thresh", 2048, [1500], 3, 8),
]


def nms_prediction(rows, n_cands, num_classes, seed, size=1024.0, conf_thres=0.5):
    """[B,rows,5+C] float32 (cx,cy,w,h,conf,cls...).  Exactly n_cands[b] rows have conf>=conf_thres;
    candidates come in spatial clusters so the merge path is exercised; all confidences distinct."""
    rng = np.random.Generator(np.random.PCG64(seed))
    B = len(n_cands)
    pred = np.zeros((B, rows, 5 + num_classes), np.float32)
    for b, n in enumerate(n_cands):
        pred[b, :, 0:2] = rng.uniform(0, size, (rows, 2))
        pred[b, :, 2:4] = rng.uniform(8, 120, (rows, 2))
        pred[b, :, 4] = rng.uniform(0.0, conf_thres * 0.98, rows)
        pred[b, :, 5:] = rng.uniform(0.01, 0.99, (rows, num_classes))
        if n == 0:
            continue
        idx = rng.permutation(rows)[:n]
        n_clusters = max(1, n // 3)
        centers = rng.uniform(60, size - 60, (n_clusters, 2))
        sizes = rng.uniform(20, 110, (n_clusters, 2))
        cls_of = rng.integers(0, num_classes, n_clusters)
        which = rng.integers(0, n_clusters, n)
        pred[b, idx, 0:2] = centers[which] + rng.normal(0, 4.0, (n, 2))
        pred[b, idx, 2:4] = sizes[which] * rng.uniform(0.9, 1.1, (n, 2))
        confs = np.linspace(conf_thres + 0.003, 0.999, n, dtype=np.float64)
        pred[b, idx, 4] = rng.permutation(confs).astype(np.float32)
        # dominant class follows the cluster most of the time, sometimes not (cross-class overlap)
        dom = np.where(rng.uniform(size=n) < 0.8, cls_of[which], rng.integers(0, num_classes, n))
        pred[b, idx, 5:] = rng.uniform(0.01, 0.45, (n, num_classes))
        pred[b, idx, 5 + dom] = rng.uniform(0.5, 0.99, n)
    return pred


def nms_dups_prediction(seed=7, num_classes=3):
    """Exact duplicate boxes (same and different class) + touching boxes: IoU == 1 and borderline cases."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = 128
    pred = np.zeros((2, rows, 5 + num_classes), np.float32)
    pred[:, :, 0:2] = rng.uniform(0, 1024, (2, rows, 2))
    pred[:, :, 2:4] = rng.uniform(8, 64, (2, rows, 2))
    pred[:, :, 4] = rng.uniform(0.0, 0.4, (2, rows))
    pred[:, :, 5:] = rng.uniform(0.01, 0.99, (2, rows, num_classes))
    for b in range(2):
        base = np.array([300.0 + 200 * b, 400.0, 50.0, 40.0], np.float32)
        for j in range(24):
            r = 5 * j + b
            pred[b, r, :4] = base if j % 3 else base + np.float32([j, 0, 0, j])
            pred[b, r, 4] = 0.5 + 0.02 * j
            pred[b, r, 5:] = 0.1
            pred[b, r, 5 + (j % num_classes if j % 2 else 0)] = 0.6 + 0.01 * j
    return pred


def nms_case_inputs(name):
    for n, rows, cands, C, seed in NMS_CASES:
        if n == name:
            if name == "dups_crossclass":
                return nms_dups_prediction(seed, C), 0.5, 0.4
            if name == "conf_low_thresh":
                return nms_prediction(rows, cands, C, seed, conf_thres=0.05), 0.05, 0.5
            return nms_prediction(rows, cands, C, seed), 0.5, 0.4
    raise KeyError(name)


MODEL_CASES = [
    # name, classes, S, B, tile start index
    ("c2_s64_b2", 2, 64, 2, 0),
    ("c3_s96_b2", 3, 96, 2, 2),
    ("c2_s160_b1", 2, 160, 1, 4),
    ("c2_s416_b1", 2, 416, 1, 5),
    ("c3_s1024_b1", 3, 1024, 1, 6),
]

TRAIN_CASES = [
    # name, classes, S, B, target seed
    ("train_c3_s128_b2", 3, 128, 2, 21),
    ("train_c2_s96_b3", 2, 96, 3, 22),
]


def model_inputs(S, B, start):
    from amyloid_yolo_paper_amd.synth import synth_tiles
    return synth_tiles(B, S, start)


def train_targets(B, C, S, seed):
    from amyloid_yolo_paper_amd.synth import synth_targets
    return synth_targets(B, C, seed=seed, max_per_tile=4, min_per_tile=2, wh_range=(0.08, 0.5), grid=S // 8)


def iou_inputs(seed=3, n=257):
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.uniform(0, 500, (n, 2)).astype(np.float32)
    b1 = np.concatenate([a, a + rng.uniform(1, 200, (n, 2)).astype(np.float32)], 1)
    c = a + rng.normal(0, 30, (n, 2)).astype(np.float32)
    b2 = np.concatenate([c, c + rng.uniform(1, 200, (n, 2)).astype(np.float32)], 1)
    return b1.astype(np.float32), b2.astype(np.float32)


def merge_inputs():
    """Inputs of the union-merge cases: name -> float32 [n,7] rows (x1,y1,x2,y2,conf,cls_conf,cls_pred)."""
    rng = np.random.Generator(np.random.PCG64(77))
    out = {}
    out["pair_overlap"] = np.array([[10.5, 10.2, 50.9, 40.1, .9, .8, 1], [30.0, 20.0, 80.0, 60.0, .7, .95, 1]], np.float32)
    out["pair_disjoint"] = np.array([[10, 10, 20, 20, .9, .8, 1], [20.5, 10, 30, 20, .7, .9, 1]], np.float32)
    out["touching_edge"] = np.array([[10, 10, 20, 20, .9, .8, 0], [20, 10, 30, 20, .7, .9, 0]], np.float32)
    out["cross_class"] = np.array([[10, 10, 50, 50, .9, .8, 1], [20, 20, 60, 60, .7, .9, 0], [25, 25, 70, 70, .6, .9, 2],
                                   [30, 30, 75, 75, .65, .9, 2]], np.float32)
    out["chain3"] = np.array([[10, 10, 40, 40, .9, .8, 1], [35, 35, 70, 70, .8, .7, 1], [65, 65, 100, 100, .7, .6, 1]], np.float32)
    out["degenerate"] = np.array([[10.2, 10.2, 10.9, 30.0, .9, .8, 1], [5, 5, 40, 40, .8, .7, 1], [100, 100, 99, 120, .5, .5, 0]], np.float32)
    out["single"] = np.array([[1.5, 2.5, 30.5, 40.5, .9, .8, 0]], np.float32)
    n = 40
    xy = rng.uniform(0, 900, (n, 2))
    wh = rng.uniform(10, 160, (n, 2))
    det = np.concatenate([xy, xy + wh, rng.uniform(0.5, 1, (n, 2)), rng.integers(0, 2, (n, 1))], 1).astype(np.float32)
    out["random40"] = det
    n = 120
    xy = rng.uniform(0, 1500, (n, 2))
    wh = rng.uniform(8, 120, (n, 2))
    out["random120_3class"] = np.concatenate([xy, xy + wh, rng.uniform(0.5, 1, (n, 2)), rng.integers(0, 3, (n, 1))], 1).astype(np.float32)
    return out


def stats_inputs(seed=55, B=12, C=3, size=416.0):
    """Evaluation-statistics case: per image a list of [n,7] detections (or None) in descending-score order
    and targets [nT,6] = (sample, class, x1, y1, x2, y2) in pixels; detections are jittered copies of some targets (true
    positives at various IoUs, duplicates of the same target, wrong classes) plus random boxes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    outputs, targets = [], []
    for b in range(B):
        nt = int(rng.integers(1, 14)) if b != 1 else 0          # image 1 has no targets
        xy = rng.uniform(0, size - 80, (nt, 2))
        wh = rng.uniform(20, 80, (nt, 2))
        cls = rng.integers(0, C, nt)
        tb = np.concatenate([xy, xy + wh], 1)
        for k in range(nt):
            targets.append([b, cls[k], *tb[k]])
        if b == 3:
            outputs.append(None)                                 # image 3 has no detections
            continue
        rows = []
        for k in range(nt):
            for _ in range(int(rng.integers(0, 4))):             # 0..3 detections near each target
                jit = rng.normal(0, rng.choice([1.0, 3.0, 8.0, 25.0]), 4)
                c = cls[k] if rng.uniform() < 0.8 else (cls[k] + 1) % C
                rows.append([*(tb[k] + jit), rng.uniform(0.5, 1), rng.uniform(0.5, 1), c])
        for _ in range(int(rng.integers(1, 6))):                 # unrelated boxes
            p = rng.uniform(0, size - 60, 2)
            rows.append([*p, *(p + rng.uniform(15, 60, 2)), rng.uniform(0.5, 1), rng.uniform(0.5, 1), rng.integers(0, C)])
        rows = np.asarray(rows, np.float32)
        rows = rows[np.argsort(-(rows[:, 4] * rows[:, 5]), kind="stable")]
        outputs.append(rows)
    return outputs, np.asarray(targets, np.float32).reshape(-1, 6)
