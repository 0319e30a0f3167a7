"""Detection statistics (reference ``utils/utils.py:69-190``): greedy true-positive matching on the device
(``ay_match_detections``), VOC-style AP per class on the host (a sort and two cumulative sums over the detections)."""
import numpy as np
import torch



def get_batch_statistics(outputs, targets, iou_threshold):
    """Reference ``utils/utils.py:154-190``: outputs = list of [n,7] | None (what ``non_max_suppression`` returns), targets
    [nT,6] = (sample, class, x1, y1, x2, y2 in pixels) -> [[true_positives, scores, labels]] per image with detections.
    The greedy matching of the whole batch runs in one device kernel (``ay_match_detections``, one wavefront per image)."""
    import ctypes as C

    from . import _lib
    from ._lib import check, ptr
    from .utils import _to_dev
    B = len(outputs)
    present = [i for i, o in enumerate(outputs) if o is not None]
    if not present:
        return []
    max_det = max(int(outputs[i].shape[0]) for i in present)
    max_det = max(max_det, 1)
    dev = _to_dev(torch.zeros(1)).device
    rows = torch.zeros(B, max_det, 7, device=dev, dtype=torch.float32)
    count = torch.zeros(B, device=dev, dtype=torch.int32)
    for i in present:
        n = int(outputs[i].shape[0])
        rows[i, :n] = outputs[i].detach().to(device=dev, dtype=torch.float32)
        count[i] = n
    tg = torch.as_tensor(targets, dtype=torch.float32).to(dev).contiguous().reshape(-1, 6)
    tp = torch.empty(B, max_det, device=dev, dtype=torch.float32)
    ovf = torch.zeros(1, device=dev, dtype=torch.int32)
    nT = int(tg.shape[0])
    check(_lib.lib().ay_match_detections(ptr(rows), ptr(count), B, max_det, ptr(tg) if nT else None, nT, C.c_float(float(iou_threshold)),
                                         ptr(tp), ptr(ovf), _lib.stream_ptr()), "ay_match_detections")
    tp = tp.cpu().numpy()
    if int(ovf.item()):
        raise _lib.AyError("ay_match_detections: an image has more than 2048 targets")
    batch_metrics = []
    for i in present:
        o = outputs[i].detach().cpu()
        batch_metrics.append([tp[i, : o.shape[0]].astype(np.float64), o[:, 4], o[:, -1]])
    return batch_metrics


def _class_segments(keys):
    """sorted integer keys -> (unique keys, start index of each run, one-past-the-end index of each run)"""
    cut = np.flatnonzero(np.diff(keys)) + 1
    starts = np.concatenate(([0], cut))
    return keys[starts], starts, np.concatenate((cut, [keys.size]))


def average_precision(true_pos, n_truth):
    """Area under the monotone precision envelope of ONE class (the reference's VOC-style ``compute_ap``,
    ``utils/utils.py:126-150``), from the true-positive flags of its detections in descending-confidence order.
    Returns (ap, final precision, final recall).  The curve is walked once from the right: the envelope value at a
    detection is the best precision at or after it, and the area grows by (recall step) x (envelope) wherever the recall
    moves -- i.e. at the true positives -- plus the closing step to recall 1 at precision 0, which adds nothing."""
    hits = np.cumsum(true_pos)
    false_alarms = np.cumsum(1.0 - true_pos)
    precision = hits / (hits + false_alarms)
    recall = hits / (n_truth + 1e-16)
    envelope = np.maximum.accumulate(precision[::-1])[::-1]
    step = np.diff(np.concatenate(([0.0], recall)))
    moved = step != 0
    return float(np.sum(step[moved] * envelope[moved])), float(precision[-1]), float(recall[-1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """Reference ``utils/utils.py:71-123``: precision, recall, AP, F1 per ground-truth class (and the class ids, int32), from
    the flat per-detection arrays of a whole evaluation.  One stable sort by (class, descending confidence) lays every
    class's detections out as a contiguous run; classes that occur only among the detections are ignored, a
    ground-truth class without detections scores zero."""
    tp = np.asarray(tp, np.float64)
    conf = np.asarray(conf)
    pred_cls = np.asarray(pred_cls)
    truth_ids, truth_counts = np.unique(np.asarray(target_cls), return_counts=True)
    order = np.lexsort((-conf, pred_cls))           # primary key: class, secondary: confidence descending (stable)
    tp, pred_sorted = tp[order], pred_cls[order]
    runs = {}
    if pred_sorted.size:
        ids, lo, hi = _class_segments(pred_sorted)
        runs = {c: (a, b) for c, a, b in zip(ids.tolist(), lo.tolist(), hi.tolist())}
    prec, rec, ap = (np.zeros(truth_ids.size) for _ in range(3))
    for k, (c, n_truth) in enumerate(zip(truth_ids.tolist(), truth_counts.tolist())):
        if c in runs:
            a, b = runs[c]
            ap[k], prec[k], rec[k] = average_precision(tp[a:b], n_truth)
    f1 = 2 * prec * rec / (prec + rec + 1e-16)
    return prec, rec, ap, f1, truth_ids.astype("int32")
