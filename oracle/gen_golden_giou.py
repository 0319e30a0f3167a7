"""Writes tests/golden/giou_kat.json: closed-form GIoU values (exact rationals) of hand-picked integer boxes, from the published
definition (Rezatofighi et al., CVPR 2019): GIoU = |A∩B|/|A∪B| − |C∖(A∪B)|/|C| with C the smallest enclosing box.  The reference
repository has no GIoU, so these vectors -- not an implementation -- are what pins the oracle's and the kernels' GIoU.
TEST INFRASTRUCTURE ONLY."""
import fractions as F
import json
import os

CASES = [
    ("identical", [0, 0, 2, 3], [0, 0, 2, 3]), ("contained_centre", [0, 0, 4, 4], [1, 1, 3, 3]), ("contained_corner", [0, 0, 4, 4], [0, 0, 2, 2]),
    ("partial_overlap_diagonal", [0, 0, 2, 2], [1, 1, 3, 3]), ("partial_overlap_side", [0, 0, 4, 2], [2, 0, 6, 2]),
    ("touching_edge", [0, 0, 1, 1], [1, 0, 2, 1]), ("touching_corner", [0, 0, 1, 1], [1, 1, 2, 2]),
    ("disjoint_side_by_side", [0, 0, 1, 1], [2, 0, 3, 1]), ("disjoint_diagonal", [0, 0, 1, 1], [3, 3, 4, 4]), ("far_apart", [0, 0, 1, 1], [99, 0, 100, 1]),
    ("cross", [1, 0, 2, 3], [0, 1, 3, 2]), ("thin_vs_square", [0, 0, 10, 1], [0, 0, 1, 10]),
]


def main():
    out = []
    for name, a, b in CASES:
        iw = max(0, min(a[2], b[2]) - max(a[0], b[0]))
        ih = max(0, min(a[3], b[3]) - max(a[1], b[1]))
        inter = F.Fraction(iw * ih)
        union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
        hull = F.Fraction((max(a[2], b[2]) - min(a[0], b[0])) * (max(a[3], b[3]) - min(a[1], b[1])))
        g = inter / union - (hull - union) / hull
        out.append({"name": name, "box1": a, "box2": b, "iou": [int(inter), int(union)], "giou": [g.numerator, g.denominator], "giou_float": float(g)})
    doc = {"source": "Closed-form values of GIoU(A,B) = |A∩B|/|A∪B| − |C∖(A∪B)|/|C| (C = smallest enclosing box), the definition of "
                     "Rezatofighi et al., 'Generalized Intersection over Union' (CVPR 2019), eq. (2)-(3) / Algorithm 1, on integer-coordinate "
                     "boxes; iou = [|A∩B|, |A∪B|], giou = [numerator, denominator] as exact rationals (computed with Python fractions by "
                     "oracle/gen_golden_giou.py, no implementation under test involved).", "cases": out}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "giou_kat.json")
    json.dump(doc, open(path, "w"), indent=1, ensure_ascii=False)


if __name__ == "__main__":
    main()
