"""CPU oracle for the YOLOv3 hot path -- TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm (keiserlab/amyloid-yolo-paper:
``models.py``, ``utils/utils.py``) on the CPU, each function citing the
reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.
The product path (``amyloid_yolo_paper_amd``) never imports this package and
fails loudly when the HIP library is missing.

Pinning: the reference's own tests hold no numeric vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, imported in the build container by ``oracle/gen_golden.py`` and
committed as fixtures under ``tests/golden/`` (``tests/test_oracle_golden.py``
checks every one of them on the CPU).  GIoU has no reference implementation
(SURVEY F3): that one function is "parity unpinned".
"""
