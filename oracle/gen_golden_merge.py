"""Golden vectors for the union-merge post-processing, generated from the reference itself.

`core.py` cannot be imported here (cv2 / skimage / torchvision are not installed and the module does file-system work at
import), so the two pure functions are taken out of its source with `ast` and executed with torch alone:
`combineIfOverlapping` (core.py:326-364) and `mergeDetections` (core.py:366-423).  Only inputs and outputs are stored
(tests/golden/merge_cases.npz); the reference's text is not.  Run in the build container:

    python oracle/gen_golden_merge.py
"""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference/core.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "merge_cases.npz")


def load_reference_functions():
    tree = ast.parse(open(REF).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("combineIfOverlapping", "mergeDetections")]
    assert len(keep) == 2
    ns = {"torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns["mergeDetections"]


def cases():
    """name -> float32 [n,7] detections (x1,y1,x2,y2,conf,cls_conf,cls_pred)"""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), ".."))
    import golden_cases as gc
    return gc.merge_inputs()


if __name__ == "__main__":
    merge = load_reference_functions()
    out = {}
    for name, det in cases().items():
        res = merge(torch.from_numpy(det))
        res = res.reshape(-1, 7) if res.numel() else res.reshape(0, 7)
        out[name] = res.numpy().astype(np.float64)  # merged corners are Python ints, confidences float32 values
        print(name, det.shape, "->", tuple(res.shape))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)
