"""ay_plan_* host logic without a GPU: the lowering of the cfg graph (Darknet._lower) and the arena layout of
ay_plan_create (csrc/ay_plan.hip).  No kernel is launched: weights are host tensors standing in for device pointers."""
import ctypes as C
import os

import pytest
import torch

from amyloid_yolo_paper_amd import _lib, cfg_gen
from amyloid_yolo_paper_amd._lib import ConvDesc, PlanOp
from amyloid_yolo_paper_amd.models import Darknet, _pad_to


def fake_prep(m):
    """what Darknet._prepare builds, with 1-element host tensors (only their addresses enter the plan)"""
    prep = {"layers": {}}
    for i, e in enumerate(m._graph):
        if e["type"] != "convolutional":
            continue
        first = i == 0 and e["cin"] == 3 and e["k"] == 3 and e["stride"] == 1 and e["cout"] == 32
        t = lambda: torch.zeros(1)
        entry = dict(scale=t(), shift=t(), cpad=_pad_to(e["cout"], 32), w=t(), stem=first)
        if first:
            entry["w0_bf16"] = t()
        else:
            entry["packed"] = t()
        prep["layers"][i] = entry
    return prep


def reads_of(o):
    r = []
    if o.kind not in (_lib.OP_STEM_S2_FUSED, _lib.OP_STEM):
        r.append(o.src)
    if o.kind in (_lib.OP_CONV1X1_CAT, _lib.OP_CONCAT_UPSAMPLE) and o.src2 != _lib.PLAN_NONE:
        r.append(o.src2)
    if o.kind == _lib.OP_CONV and o.res != _lib.PLAN_NONE:
        r.append(o.res)
    return r


def create(ops, vbytes, S, N, dtype=0):
    L = _lib.lib()
    arr = (PlanOp * len(ops))(*ops)
    vb = (C.c_size_t * len(vbytes))(*vbytes)
    h = C.c_void_p()
    rc = L.ay_plan_create(arr, len(ops), vb, len(vbytes), S, N, dtype, C.byref(h))
    return rc, h


@pytest.mark.parametrize("opts", [dict(), dict(fuse_blocks=False, fold_routes=False), dict(stem_mode="fp32"), dict(precision="fp16")], ids=str)
def test_lowering_and_arena(tmp_cfg_dir, opts):
    L = _lib.lib()
    m = Darknet(cfg_gen.write_cfg(3, tmp_cfg_dir))
    m.precision = "bf16"
    for k, v in opts.items():
        setattr(m, k, v)
    B, S = 4, 416
    ops, vbytes = m._lower(B, S, fake_prep(m))
    kinds = [o.kind for o in ops]
    assert kinds.count(_lib.OP_DECODE) == 3
    if not opts or "precision" in opts:   # the half-precision plan is the same op list on the other storage type
        assert kinds[0] == _lib.OP_STEM_S2_FUSED and kinds.count(_lib.OP_RESBLOCK) == 1 and kinds.count(_lib.OP_CONV1X1_CAT) == 2
        assert kinds.count(_lib.OP_CONCAT_UPSAMPLE) == 0          # both routes ride the 1x1 loader
        assert len(ops) == 75 - 1 - 1 + 3                         # 75 convs; the stem pair and one block are one op each; 3 decodes
    if opts.get("fold_routes") is False:
        assert kinds.count(_lib.OP_CONCAT_UPSAMPLE) == 2 and kinds.count(_lib.OP_RESBLOCK) == 0
    if opts.get("stem_mode") == "fp32":
        assert kinds[0] == _lib.OP_STEM
    rc, h = create(ops, vbytes, S, m.num_boxes(S), int(m.precision == "fp16"))
    assert rc == 0, L.ay_last_error()
    try:
        arena = L.ay_plan_workspace_bytes(h)
        off = [L.ay_plan_value_offset(h, v) for v in range(len(vbytes))]
        assert arena < 0.25 * sum(vbytes)                          # lifetimes are short: a few layers live at a time
        assert arena >= max(vbytes) and all(o % 256 == 0 for o in off)
        # no two values whose lifetimes overlap share bytes
        born, last = {}, {}
        for i, o in enumerate(ops):
            for v in reads_of(o):
                assert v in born, (i, v)
                last[v] = i
            if o.kind != _lib.OP_DECODE:
                born[o.dst] = i
                last.setdefault(o.dst, i)
        vals = sorted(born)
        for a in vals:
            for b in vals:
                if a < b and born[a] <= last[b] and born[b] <= last[a]:
                    assert off[a] + vbytes[a] <= off[b] or off[b] + vbytes[b] <= off[a], (a, b)
    finally:
        L.ay_plan_destroy(h)


def test_plan_rejects_broken_dataflow():
    L = _lib.lib()

    def conv(src, dst, res=_lib.PLAN_NONE):
        o = PlanOp()
        o.kind, o.src, o.src2, o.res, o.dst = _lib.OP_CONV, src, _lib.PLAN_NONE, res, dst
        o.conv = ConvDesc(1, 16, 32, 8, 8, 8, 8, 1, 1, 1, 0, 32)
        return o

    stem = PlanOp()
    stem.kind, stem.src, stem.src2, stem.res, stem.dst = _lib.OP_STEM, _lib.PLAN_NONE, _lib.PLAN_NONE, _lib.PLAN_NONE, 0
    for ops, what in (([stem, conv(1, 2)], b"before it is written"), ([stem, conv(0, 0)], b"written twice"),
                      ([stem, conv(0, 1), conv(0, 1)], b"written twice"), ([stem, conv(0, 7)], b"out of range"),
                      ([stem, conv(0, 1, res=2)], b"before it is written")):
        rc, h = create(ops, [64, 64, 64], 32, 3)
        assert rc == -1 and what in L.ay_last_error(), (what, L.ay_last_error())
    rc, h = create([stem, conv(0, 1)], [64, 64], 32, 3, dtype=5)
    assert rc == -1 and b"dtype" in L.ay_last_error()
    rc, h = create([stem, conv(0, 1), conv(1, 2, res=0)], [1000, 300, 300], 32, 3)
    assert rc == 0
    assert L.ay_plan_workspace_bytes(h) == 1024 + 512 + 512          # all three alive at the last op
    L.ay_plan_destroy(h)
    rc, h = create([stem, conv(0, 1), conv(1, 2)], [1000, 300, 300], 32, 3)
    assert rc == 0 and L.ay_plan_value_offset(h, 2) == 0              # value 0 is dead when value 2 is born
    assert L.ay_plan_workspace_bytes(h) == 1024 + 512
    L.ay_plan_destroy(h)
