"""NumPy fp32 restatement of the reference's box math (TEST INFRASTRUCTURE ONLY).

Every function names the reference lines it follows (paths relative to the
reference repo).  All arithmetic is done in float32 in the reference's operation
order so that threshold comparisons (``>=conf_thres``, ``>nms_thres``,
``>ignore_thres``) land on the same side as in the reference.
"""
import numpy as np

F32 = np.float32
EPS = F32(1e-16)


def xywh2xyxy(x):
    """utils/utils.py:53-59 -- (cx,cy,w,h) -> (x1,y1,x2,y2)."""
    x = np.asarray(x, F32)
    y = np.empty_like(x)
    half_w = x[..., 2] / F32(2)
    half_h = x[..., 3] / F32(2)
    y[..., 0] = x[..., 0] - half_w
    y[..., 1] = x[..., 1] - half_h
    y[..., 2] = x[..., 0] + half_w
    y[..., 3] = x[..., 1] + half_h
    return y


def bbox_wh_iou(wh1, wh2):
    """utils/utils.py:193-199 -- IoU of boxes sharing a corner; wh1 [2], wh2 [n,2] -> [n]."""
    wh1 = np.asarray(wh1, F32)
    wh2 = np.asarray(wh2, F32).reshape(-1, 2)
    w1, h1 = wh1[0], wh1[1]
    w2, h2 = wh2[:, 0], wh2[:, 1]
    inter = np.minimum(w1, w2) * np.minimum(h1, h2)
    union = (w1 * h1 + EPS) + w2 * h2 - inter
    return (inter / union).astype(F32)


def bbox_iou(box1, box2, x1y1x2y2=True):
    """utils/utils.py:202-232 -- +1-pixel IoU, broadcast [1|n,4] x [n,4] -> [n]."""
    box1 = np.asarray(box1, F32).reshape(-1, 4)
    box2 = np.asarray(box2, F32).reshape(-1, 4)
    if not x1y1x2y2:
        b1_x1, b1_x2 = box1[:, 0] - box1[:, 2] / F32(2), box1[:, 0] + box1[:, 2] / F32(2)
        b1_y1, b1_y2 = box1[:, 1] - box1[:, 3] / F32(2), box1[:, 1] + box1[:, 3] / F32(2)
        b2_x1, b2_x2 = box2[:, 0] - box2[:, 2] / F32(2), box2[:, 0] + box2[:, 2] / F32(2)
        b2_y1, b2_y2 = box2[:, 1] - box2[:, 3] / F32(2), box2[:, 1] + box2[:, 3] / F32(2)
    else:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1[:, 0], box1[:, 1], box1[:, 2], box1[:, 3]
        b2_x1, b2_y1, b2_x2, b2_y2 = box2[:, 0], box2[:, 1], box2[:, 2], box2[:, 3]
    ix1 = np.maximum(b1_x1, b2_x1)
    iy1 = np.maximum(b1_y1, b2_y1)
    ix2 = np.minimum(b1_x2, b2_x2)
    iy2 = np.minimum(b1_y2, b2_y2)
    one = F32(1)
    inter = np.maximum(ix2 - ix1 + one, F32(0)) * np.maximum(iy2 - iy1 + one, F32(0))
    a1 = (b1_x2 - b1_x1 + one) * (b1_y2 - b1_y1 + one)
    a2 = (b2_x2 - b2_x1 + one) * (b2_y2 - b2_y1 + one)
    return (inter / (a1 + a2 - inter + EPS)).astype(F32)


def bbox_giou(box1, box2):
    """GIoU, corner boxes, no +1 rule.  The reference has no GIoU: this function is pinned by the closed-form vectors
    of the published definition instead (tests/golden/giou_kat.json, oracle/gen_golden_giou.py; tests/test_oracle_golden.py)."""
    box1 = np.asarray(box1, F32).reshape(-1, 4)
    box2 = np.asarray(box2, F32).reshape(-1, 4)
    iw = np.maximum(np.minimum(box1[:, 2], box2[:, 2]) - np.maximum(box1[:, 0], box2[:, 0]), F32(0))
    ih = np.maximum(np.minimum(box1[:, 3], box2[:, 3]) - np.maximum(box1[:, 1], box2[:, 1]), F32(0))
    inter = iw * ih
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    union = a1 + a2 - inter + EPS
    cw = np.maximum(box1[:, 2], box2[:, 2]) - np.minimum(box1[:, 0], box2[:, 0])
    ch = np.maximum(box1[:, 3], box2[:, 3]) - np.minimum(box1[:, 1], box2[:, 1])
    hull = cw * ch + EPS
    return (inter / union - (hull - union) / hull).astype(F32)


def rescale_boxes(boxes, current_dim, original_shape):
    """utils/utils.py:36-50 -- undo pad-to-square + resize (note the float floor-division)."""
    boxes = np.array(boxes, F32, copy=True)
    orig_h, orig_w = original_shape
    pad_x = max(orig_h - orig_w, 0) * (current_dim / max(original_shape))
    pad_y = max(orig_w - orig_h, 0) * (current_dim / max(original_shape))
    unpad_h = current_dim - pad_y
    unpad_w = current_dim - pad_x
    boxes[:, 0] = ((boxes[:, 0] - pad_x // 2) / unpad_w) * orig_w
    boxes[:, 1] = ((boxes[:, 1] - pad_y // 2) / unpad_h) * orig_h
    boxes[:, 2] = ((boxes[:, 2] - pad_x // 2) / unpad_w) * orig_w
    boxes[:, 3] = ((boxes[:, 3] - pad_y // 2) / unpad_h) * orig_h
    return boxes


def nms_merge_image(image_pred, conf_thres=0.5, nms_thres=0.4):
    """One image of utils/utils.py:246-271.

    ``image_pred`` [N,5+C] with boxes ALREADY as corners.  Returns
    ``(rows [n,7] f32, keep_idx [n] int64, clusters list[int64 array])`` or
    ``(None, empty, [])``.  ``keep_idx`` are original row numbers of the cluster
    heads, ``clusters[i]`` the original rows merged into head i (head first).
    Sort ties (unspecified in the reference, ``:255``) break towards the lower row.
    """
    image_pred = np.asarray(image_pred, F32)
    cand = np.nonzero(image_pred[:, 4] >= F32(conf_thres))[0]
    if cand.size == 0:
        return None, np.zeros(0, np.int64), []
    p = image_pred[cand]
    score = p[:, 4] * p[:, 5:].max(1)
    order = np.argsort(-score, kind="stable")
    p, cand = p[order], cand[order]
    cls_conf = p[:, 5:].max(1)
    cls_pred = p[:, 5:].argmax(1).astype(F32)
    det = np.concatenate([p[:, :5], cls_conf[:, None], cls_pred[:, None]], 1).astype(F32)
    alive = np.ones(det.shape[0], bool)
    rows, keep, clusters = [], [], []
    thr = F32(nms_thres)
    for i in range(det.shape[0]):
        if not alive[i]:
            continue
        rem = np.nonzero(alive)[0]  # rem[0] == i
        iou = bbox_iou(det[i:i + 1, :4], det[rem, :4])
        invalid = (iou > thr) & (det[rem, 6] == det[i, 6])
        members = rem[invalid]
        w = det[members, 4:5]
        head = det[i].copy()
        head[:4] = (w * det[members, :4]).sum(0, dtype=F32) / w.sum(dtype=F32)
        rows.append(head)
        keep.append(cand[i])
        clusters.append(cand[members].astype(np.int64))
        alive[members] = False
    return np.stack(rows).astype(F32), np.asarray(keep, np.int64), clusters


def non_max_suppression(prediction, conf_thres=0.5, nms_thres=0.4):
    """utils/utils.py:235-273 -- returns (list of rows|None, list of keep_idx, list of clusters).

    Like the reference it converts ``prediction[..., :4]`` to corners IN PLACE.
    """
    prediction[..., :4] = xywh2xyxy(prediction[..., :4])
    outs, keeps, clusters = [], [], []
    for image_pred in prediction:
        r, k, c = nms_merge_image(image_pred, conf_thres, nms_thres)
        outs.append(r)
        keeps.append(k)
        clusters.append(c)
    return outs, keeps, clusters


def decode(head, anchors, num_classes, img_dim):
    """models.py:137-169 -- head [B,A*(5+C),G,G] -> (output [B,A*G*G,5+C], pred_boxes [B,A,G,G,4] in grid units,
    and the sigmoid/raw pieces the loss needs)."""
    head = np.asarray(head, F32)
    B, _, G, _ = head.shape
    A = len(anchors)
    p = head.reshape(B, A, 5 + num_classes, G, G).transpose(0, 1, 3, 4, 2)
    sig = lambda v: (F32(1) / (F32(1) + np.exp(-v, dtype=F32))).astype(F32)
    x, y = sig(p[..., 0]), sig(p[..., 1])
    w, h = p[..., 2], p[..., 3]
    conf, cls = sig(p[..., 4]), sig(p[..., 5:])
    stride = F32(img_dim / G)
    gx = np.arange(G, dtype=F32).reshape(1, 1, 1, G)
    gy = np.arange(G, dtype=F32).reshape(1, 1, G, 1)
    sa = np.asarray([(aw / stride, ah / stride) for aw, ah in anchors], F32)
    boxes = np.empty(p[..., :4].shape, F32)
    boxes[..., 0] = x + gx
    boxes[..., 1] = y + gy
    boxes[..., 2] = np.exp(w, dtype=F32) * sa[:, 0].reshape(1, A, 1, 1)
    boxes[..., 3] = np.exp(h, dtype=F32) * sa[:, 1].reshape(1, A, 1, 1)
    out = np.concatenate([boxes.reshape(B, -1, 4) * stride, conf.reshape(B, -1, 1), cls.reshape(B, -1, num_classes)], -1)
    return out.astype(F32), boxes, dict(x=x, y=y, w=w, h=h, conf=conf, cls=cls, scaled_anchors=sa)


def build_targets(pred_boxes, pred_cls, target, anchors, ignore_thres):
    """utils/utils.py:276-330 -- same 10-tuple, same order.

    pred_boxes [B,A,G,G,4] (grid units, cxcywh), pred_cls [B,A,G,G,C], target [nT,6],
    anchors [A,2] already divided by the stride.  Duplicate (b,a,gj,gi) scatters are
    last-writer-wins, as on the reference's CPU path.
    """
    pred_boxes = np.asarray(pred_boxes, F32)
    pred_cls = np.asarray(pred_cls, F32)
    target = np.asarray(target, F32)
    anchors = np.asarray(anchors, F32)
    nB, nA, nG = pred_boxes.shape[0], pred_boxes.shape[1], pred_boxes.shape[2]
    nC = pred_cls.shape[-1]
    obj_mask = np.zeros((nB, nA, nG, nG), bool)
    noobj_mask = np.ones((nB, nA, nG, nG), bool)
    class_mask = np.zeros((nB, nA, nG, nG), F32)
    iou_scores = np.zeros((nB, nA, nG, nG), F32)
    tx = np.zeros((nB, nA, nG, nG), F32)
    ty = np.zeros((nB, nA, nG, nG), F32)
    tw = np.zeros((nB, nA, nG, nG), F32)
    th = np.zeros((nB, nA, nG, nG), F32)
    tcls = np.zeros((nB, nA, nG, nG, nC), F32)

    target_boxes = target[:, 2:6] * F32(nG)
    gxy, gwh = target_boxes[:, :2], target_boxes[:, 2:]
    ious = np.stack([bbox_wh_iou(a, gwh) for a in anchors])  # [nA,nT]
    best_n = ious.argmax(0)
    b = target[:, 0].astype(np.int64)
    labels = target[:, 1].astype(np.int64)
    gx, gy = gxy[:, 0], gxy[:, 1]
    gw, gh = gwh[:, 0], gwh[:, 1]
    gi, gj = gx.astype(np.int64), gy.astype(np.int64)  # trunc toward zero, like .long()
    obj_mask[b, best_n, gj, gi] = True
    noobj_mask[b, best_n, gj, gi] = False
    for i in range(target.shape[0]):
        noobj_mask[b[i], ious[:, i] > F32(ignore_thres), gj[i], gi[i]] = False
    tx[b, best_n, gj, gi] = gx - np.floor(gx)
    ty[b, best_n, gj, gi] = gy - np.floor(gy)
    tw[b, best_n, gj, gi] = np.log(gw / anchors[best_n][:, 0] + EPS, dtype=F32)
    th[b, best_n, gj, gi] = np.log(gh / anchors[best_n][:, 1] + EPS, dtype=F32)
    tcls[b, best_n, gj, gi, labels] = 1
    class_mask[b, best_n, gj, gi] = (pred_cls[b, best_n, gj, gi].argmax(-1) == labels).astype(F32)
    iou_scores[b, best_n, gj, gi] = bbox_iou(pred_boxes[b, best_n, gj, gi], target_boxes, x1y1x2y2=False)
    tconf = obj_mask.astype(F32)
    return iou_scores, class_mask, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf


def get_batch_statistics(outputs, targets, iou_threshold):
    """utils/utils.py:154-190 -- greedy TP matching; outputs list of [n,7]|None, targets [nT,6] (corners*img)."""
    metrics = []
    targets = np.asarray(targets, F32)
    for i, out in enumerate(outputs):
        if out is None:
            continue
        out = np.asarray(out, F32)
        boxes, scores, labels = out[:, :4], out[:, 4], out[:, -1]
        tp = np.zeros(boxes.shape[0])
        ann = targets[targets[:, 0] == i][:, 1:]
        tlabels = ann[:, 0] if len(ann) else []
        if len(ann):
            detected = []
            tboxes = ann[:, 1:]
            for pi, (pbox, plabel) in enumerate(zip(boxes, labels)):
                if len(detected) == len(ann):
                    break
                if plabel not in tlabels:
                    continue
                ious = bbox_iou(pbox[None], tboxes)
                bi = int(ious.argmax())
                if ious[bi] >= iou_threshold and bi not in detected:
                    tp[pi] = 1
                    detected.append(bi)
        metrics.append([tp, scores, labels])
    return metrics


def compute_ap(recall, precision):
    """utils/utils.py:126-151 -- VOC-style envelope AP."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """utils/utils.py:71-123."""
    tp, conf, pred_cls, target_cls = map(np.asarray, (tp, conf, pred_cls, target_cls))
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes = np.unique(target_cls)
    ap, p, r = [], [], []
    for c in unique_classes:
        i = pred_cls == c
        n_gt = (target_cls == c).sum()
        n_p = i.sum()
        if n_p == 0 and n_gt == 0:
            continue
        elif n_p == 0 or n_gt == 0:
            ap.append(0); r.append(0); p.append(0)
        else:
            fpc = (1 - tp[i]).cumsum()
            tpc = (tp[i]).cumsum()
            recall_curve = tpc / (n_gt + 1e-16)
            r.append(recall_curve[-1])
            precision_curve = tpc / (tpc + fpc)
            p.append(precision_curve[-1])
            ap.append(compute_ap(recall_curve, precision_curve))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, unique_classes.astype("int32")


def merge_detections_ordered(det):
    """``mergeDetections`` (core.py:366-423, with ``combineIfOverlapping`` core.py:326-364) with the one thing the reference leaves
    to CPython -- the iteration order of its set of row tuples -- made explicit: rows in input order, merged rows appended in
    creation order.  Everything else as the reference has it: identical rows collapse (``set``); per pass all pairs i < j of the
    pass's list; only classes 0 and 1; ``int()`` truncation of x1, y1, x2 - x1, y2 - y1; overlap = the pixel rectangles
    [x, x+w) x [y, y+h) share a pixel; merged row = (left, top, last covered column, last covered row, min conf, min cls_conf,
    label); skipped if that row is already in the set; ``removed`` is a set of row VALUES; passes until nothing changes.
    det: float32 [n,7].  Returns float64 [m,7] in list order.  (The device kernel ay_merge_detections implements exactly this.)"""
    rows, seen = [], set()
    for r in np.asarray(det, np.float32).reshape(-1, 7).tolist():
        t = tuple(r)
        if t not in seen:
            seen.add(t)
            rows.append(t)
    live = [True] * len(rows)
    removed = set()
    while True:
        changed = False
        limit = len(rows)
        for i in range(limit):
            for j in range(i + 1, limit):
                ei, ej = rows[i], rows[j]
                if not live[i] or not live[j]:
                    continue
                if not ((ei[6] == 1 == ej[6]) or (ei[6] == 0 == ej[6])):
                    continue
                if ei in removed or ej in removed:
                    continue
                xi, yi, wi, hi = int(ei[0]), int(ei[1]), int(ei[2] - ei[0]), int(ei[3] - ei[1])
                xj, yj, wj, hj = int(ej[0]), int(ej[1]), int(ej[2] - ej[0]), int(ej[3] - ej[1])
                if wi <= 0 or hi <= 0 or wj <= 0 or hj <= 0:
                    continue
                if min(xi + wi, xj + wj) <= max(xi, xj) or min(yi + hi, yj + hj) <= max(yi, yj):
                    continue
                left, top = min(xi, xj), min(yi, yj)
                right, bottom = max(xi + wi, xj + wj) - 1, max(yi + hi, yj + hj) - 1
                new = (left, top, right, bottom, min(ei[4], ej[4]), min(ei[5], ej[5]), ei[6])
                if any(live[k] and rows[k] == new for k in range(len(rows))):
                    continue
                rows.append(new)
                live.append(True)
                live[i] = live[j] = False
                removed.add(ei)
                removed.add(ej)
                changed = True
        if not changed:
            break
    return np.asarray([rows[k] for k in range(len(rows)) if live[k]], np.float64).reshape(-1, 7)
