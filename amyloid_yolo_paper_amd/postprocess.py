"""Host-side post-processing of per-tile detections: the union-merge of overlapping same-class boxes
that the paper's inference path applies after NMS (reference ``core.py:366-423`` ``mergeDetections`` with
``combineIfOverlapping`` ``core.py:326-364``; enabled by ``--merge_boxes True``, ``README.md:39``).

The reference decides "overlap" by intersecting two Python sets of integer pixel coordinates (O(w*h) per pair); here the
same decision and the same merged box come from interval arithmetic.  Everything else that shapes the result is kept as
the reference has it, because it is observable: corners and sizes are truncated with ``int()`` before the test; a merged
box is ``(left, top, right - left, bottom - top)`` of the covered PIXELS, so it is one pixel narrower and lower than the
union of the two rectangles; confidences merge by ``min``; only classes 0 and 1 take part; the pairs are visited in the
iteration order of a Python ``set`` of float tuples, an entry merged in a pass is not used again in that pass, and passes
repeat until nothing changes.  The output rows are in that set's iteration order (callers treat them as a set).

This is host code by nature (a few hundred boxes per tile, data-dependent fixed point); it is not a GPU kernel and does
not touch the device."""
import torch


def _combine_if_overlapping(b1, b2):
    """b = (x, y, w, h) integers.  (True, merged) if the pixel rectangles [x, x+w) x [y, y+h) share a pixel."""
    x1, y1, w1, h1 = b1
    x2, y2, w2, h2 = b2
    if w1 <= 0 or h1 <= 0 or w2 <= 0 or h2 <= 0:
        return False, -1  # an empty range has no pixels
    if min(x1 + w1, x2 + w2) <= max(x1, x2) or min(y1 + h1, y2 + h2) <= max(y1, y2):
        return False, -1
    left, top = min(x1, x2), min(y1, y2)
    right, bottom = max(x1 + w1, x2 + w2) - 1, max(y1 + h1, y2 + h2) - 1  # last covered pixel
    return True, (left, top, right - left, bottom - top)


def merge_detections(detections):
    """``detections``: tensor [n,7] of (x1, y1, x2, y2, conf, cls_conf, cls_pred).  Returns the merged rows as a tensor
    (same convention as the reference: ``torch.as_tensor`` of the surviving rows, in set order)."""
    entries = set(tuple(row) for row in detections.tolist())
    removed = set()
    changed = True
    while changed:
        changed = False
        order = list(entries)
        for i in range(len(order)):
            for j in range(i + 1, len(order)):
                ei, ej = order[i], order[j]
                if not ((ei[6] == 1 == ej[6]) or (ei[6] == 0 == ej[6])):
                    continue
                if ei in removed or ej in removed:
                    continue
                bi = (int(ei[0]), int(ei[1]), int(ei[2] - ei[0]), int(ei[3] - ei[1]))
                bj = (int(ej[0]), int(ej[1]), int(ej[2] - ej[0]), int(ej[3] - ej[1]))
                ok, nb = _combine_if_overlapping(bi, bj)
                if not ok:
                    continue
                merged = (nb[0], nb[1], nb[0] + nb[2], nb[1] + nb[3], min(ei[4], ej[4]), min(ei[5], ej[5]), ei[6])
                if merged not in entries:
                    entries.add(merged)
                    entries.remove(ei)
                    entries.remove(ej)
                    removed.add(ei)
                    removed.add(ej)
                    changed = True
    return torch.as_tensor([list(e) for e in entries])
