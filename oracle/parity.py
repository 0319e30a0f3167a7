"""Agreement metrics between two detection sets of the same tile (TEST INFRASTRUCTURE ONLY: tests, bench.py's cpu_baseline leg).

The reference's merge-NMS (``utils/utils.py:235-273``) emits one row per cluster head; "box indices after NMS" (BASELINE.json
north_star) are the ORIGINAL row indices (0..N-1) of those heads.  ``detection_agreement`` reports how many of the reference's
indices a second implementation reproduces and how far the matched rows lie apart."""
import numpy as np


def detection_agreement(ref_keep, ref_rows, got_keep, got_rows):
    """ref_keep/got_keep: int arrays of head row indices; ref_rows/got_rows: [n,7] (x1,y1,x2,y2,conf,cls_conf,cls_pred) or None.
    Returns dict(n_ref, n_got, matched, box_rel, dconf, cls_equal): box_rel = max over matched heads of
    max|corner difference| / max(box width, box height, 1) of the reference box; dconf = max |conf difference|."""
    ref_keep = np.asarray(ref_keep, np.int64).reshape(-1)
    got_keep = np.asarray(got_keep, np.int64).reshape(-1)
    pos = {int(k): j for j, k in enumerate(got_keep)}
    pairs = [(i, pos[int(k)]) for i, k in enumerate(ref_keep) if int(k) in pos]
    res = dict(n_ref=int(ref_keep.size), n_got=int(got_keep.size), matched=len(pairs), box_rel=0.0, dconf=0.0, cls_equal=True)
    if pairs:
        a = np.asarray(ref_rows, np.float64)[[p[0] for p in pairs]]
        b = np.asarray(got_rows, np.float64)[[p[1] for p in pairs]]
        size = np.maximum(np.maximum(a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]), 1.0)
        res["box_rel"] = float((np.abs(a[:, :4] - b[:, :4]).max(1) / size).max())
        res["dconf"] = float(np.abs(a[:, 4] - b[:, 4]).max())
        res["cls_equal"] = bool((a[:, 6] == b[:, 6]).all())
    return res


def summarize(items):
    """list of detection_agreement dicts -> totals for a bench line / an assertion"""
    n_ref = sum(d["n_ref"] for d in items)
    matched = sum(d["matched"] for d in items)
    n_got = sum(d["n_got"] for d in items)
    return dict(keep_match=round(matched / max(n_ref, 1), 4), extra_heads=round((n_got - matched) / max(n_got, 1), 4),
                max_box_rel=round(max((d["box_rel"] for d in items), default=0.0), 5),
                max_dconf=round(max((d["dconf"] for d in items), default=0.0), 5), n_ref=n_ref, n_got=n_got, n_tiles=len(items))
