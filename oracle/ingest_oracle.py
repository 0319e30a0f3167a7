"""CPU restatement of the reference's image path for ONE tile (TEST INFRASTRUCTURE ONLY -- never imported by the product).

``utils/transforms.py:96`` (torchvision ``ToTensor``: HWC uint8 -> CHW float32 / 255), ``utils/datasets.py:22-32``
(``pad_to_square``: zero padding, ``diff // 2`` first) and ``utils/datasets.py:35-37`` (``F.interpolate(mode="nearest")``),
composed from the same ATen CPU ops the reference calls.  torchvision is not installed here, so ``ToTensor`` is restated
(``img.permute(2,0,1).float().div(255)`` is its documented uint8 branch); parity for this row is pinned by ATen itself.
"""
import numpy as np
import torch
import torch.nn.functional as F


def ingest(img_u8, size, pad_value=0.0):
    """img_u8 [H,W,3] uint8 -> float32 [3,size,size]"""
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).float().div(255.0)
    _, h, w = t.shape
    diff = abs(h - w)
    p1, p2 = diff // 2, diff - diff // 2
    pad = (0, 0, p1, p2) if h <= w else (p1, p2, 0, 0)
    t = F.pad(t, pad, "constant", value=pad_value)
    return F.interpolate(t.unsqueeze(0), size=size, mode="nearest").squeeze(0)
