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


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    tp, conf, pred_cls, target_cls = (np.asarray(v) for v in (tp, conf, pred_cls, target_cls))
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes = np.unique(target_cls)
    ap, p, r = [], [], []
    for c in unique_classes:
        sel = pred_cls == c
        n_gt, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            ap.append(0); r.append(0); p.append(0)
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(), tp[sel].cumsum()
        recall_curve = tpc / (n_gt + 1e-16)
        precision_curve = tpc / (tpc + fpc)
        r.append(recall_curve[-1]); p.append(precision_curve[-1])
        ap.append(compute_ap(recall_curve, precision_curve))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, unique_classes.astype("int32")
