"""Detection statistics on the host (reference ``utils/utils.py:71-190``): greedy TP matching, VOC-style AP.
Small per-epoch bookkeeping on [n,7] rows; the IoUs come from the HIP box kernel."""
import numpy as np
import torch

from .utils import bbox_iou


def get_batch_statistics(outputs, targets, iou_threshold):
    """outputs: list of [n,7] | None; targets [nT,6] (sample, class, x1, y1, x2, y2 in pixels) -> [[tp, conf, label]]"""
    batch_metrics = []
    for sample_i, output in enumerate(outputs):
        if output is None:
            continue
        output = output.detach().cpu()
        pred_boxes, pred_scores, pred_labels = output[:, :4], output[:, 4], output[:, -1]
        true_positives = np.zeros(pred_boxes.shape[0])
        annotations = targets[targets[:, 0] == sample_i][:, 1:]
        target_labels = annotations[:, 0] if len(annotations) else []
        if len(annotations):
            detected = []
            target_boxes = annotations[:, 1:]
            for pred_i, (pred_box, pred_label) in enumerate(zip(pred_boxes, pred_labels)):
                if len(detected) == len(annotations):
                    break
                if pred_label not in target_labels:
                    continue
                iou, box_index = bbox_iou(pred_box.unsqueeze(0), target_boxes).max(0)
                if iou >= iou_threshold and box_index not in detected:
                    true_positives[pred_i] = 1
                    detected += [box_index]
        batch_metrics.append([true_positives, pred_scores, pred_labels])
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
