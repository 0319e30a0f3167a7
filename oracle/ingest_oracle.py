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


def region_tiles(raster_u8, tile, size, shrink=1):
    """WSI -> tiles, CPU restatement of what ``crop.py:13-25,44-47`` + the detect-time image path produce,
    minus the two steps this row exists to remove or cannot restate here: the JPEG Q=90 round trip through the disk, and pyvips'
    lanczos3 ``resize(0.5)`` (pyvips is not installed; the 40x -> 20x halving is a 2x2 mean, round half up).  PARITY UNPINNED for
    shrink=2; shrink=1 is pinned by construction: dzsave(layout='google') cuts the slide on a ``tile`` grid and pads edge tiles
    with the background 255, after which every tile goes through :func:`ingest`.

    raster_u8 [H,W,3] uint8 -> float32 [tiles_y*tiles_x, 3, size, size], (tiles_y, tiles_x)"""
    r = np.asarray(raster_u8)
    if shrink == 2:
        h2, w2 = r.shape[0] // 2, r.shape[1] // 2
        q = r[: 2 * h2, : 2 * w2].astype(np.uint16)
        r = ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    H, W = r.shape[:2]
    ty, tx = -(-H // tile), -(-W // tile)
    out = []
    for j in range(ty):
        for i in range(tx):
            t = np.full((tile, tile, 3), 255, np.uint8)
            c = r[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile]
            t[: c.shape[0], : c.shape[1]] = c
            out.append(ingest(t, size))
    return torch.stack(out), (ty, tx)
