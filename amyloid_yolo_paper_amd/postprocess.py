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

``merge_detections`` is the host form, identical to the reference row for row INCLUDING the order CPython's set gives them.
``merge_detections_device`` runs the same algorithm on the GPU for a whole batch (``ay_merge_detections``: one wavefront per
image) with the pair order made explicit (rows in input order, merged rows appended): the same rows as the reference whenever
the clusters are pairs, the reference's algorithm under that order for chains of merges (whose edges depend on the order
through the one-pixel shrink per merge)."""
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


def merge_detections_batch_device(dets):
    """``detect(merge_boxes=True, merge_on_device=True)``: a list of per-image ``[n,7]`` tensors (or None) through ONE
    ``ay_merge_detections`` launch (one wavefront per image) -> the same list form.  Row order is the kernel's explicit one (input
    order, merged rows appended), where ``merge_detections`` keeps the reference's set-iteration order: equal as sets of rows unless
    a chain of merges makes the reference itself order-dependent."""
    import torch
    from . import _lib
    n = [0 if d is None else int(d.shape[0]) for d in dets]
    if not dets or max(n) == 0:
        return list(dets)
    cap = _lib.lib().ay_merge_detections_max_rows()
    if max(n) > cap:
        raise _lib.AyError(f"merge_detections_batch_device: an image holds {max(n)} rows (at most {cap})")
    M = max(n)   # a merge replaces two rows by one: an image never holds more rows than it came with
    dev = torch.device("cuda", torch.cuda.current_device())
    rows = torch.zeros(len(dets), M, 7, dtype=torch.float32)
    for b, d in enumerate(dets):
        if n[b]:
            rows[b, :n[b]] = d.detach().to(dtype=torch.float32, device="cpu")
    out, cnt = merge_detections_device(rows.to(dev), torch.tensor(n, dtype=torch.int32, device=dev))
    out, cnt = out.cpu(), cnt.cpu().tolist()
    return [None if n[b] == 0 else out[b, :cnt[b]].clone() for b in range(len(dets))]


def merge_detections_device(rows, count):
    """``rows`` [B, max_rows, 7] float32 CUDA tensor of (x1, y1, x2, y2, conf, cls_conf, cls_pred), ``count`` [B] int32 valid rows per
    image (what ``utils.nms_device`` returns, after rescaling) -> (merged rows [B, max_rows, 7], merged count [B]) on the device,
    no host synchronisation."""
    from . import _lib
    L = _lib.lib()
    assert rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 3 and rows.shape[2] == 7 and rows.is_contiguous()
    B, M, _ = rows.shape
    if M > L.ay_merge_detections_max_rows():
        raise _lib.AyError(f"merge_detections_device: {M} rows per image (at most {L.ay_merge_detections_max_rows()})")
    count = count.to(device=rows.device, dtype=torch.int32).contiguous()
    out = torch.zeros_like(rows)          # the kernel writes the first out_count[b] rows of an image: the tail reads as zeros
    out_count = torch.empty_like(count)
    _lib.check(L.ay_merge_detections(_lib.ptr(rows), _lib.ptr(count), B, M, _lib.ptr(out), _lib.ptr(out_count), _lib.stream_ptr()),
               "ay_merge_detections")
    return out, out_count
