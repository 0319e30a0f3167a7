"""WSI -> tile streaming: detection over a whole-slide raster without the disk round trip of the
reference (``crop.py:13-25`` writes 1536-px JPEG tiles with pyvips ``dzsave``, ``detect.py:59-105`` reads them back).

A slide (any ``[H, W, 3]`` uint8 array: a NumPy memmap of a decoded level, a pyvips/openslide region fetched by the caller)
is walked in full-width strips of one tile row.  A strip is one contiguous slice of the raster: it goes to the device in a
single copy from a pinned staging buffer on a copy stream (two buffers, so strip i+1 uploads while strip i computes), and
``ay_ingest_region_tiles_u8`` cuts the row of tiles out of it on the device -- dzsave's 'google' layout: a ``tile`` grid from
the top-left corner, edge tiles padded with the background 255 -- with the optional 40x -> 20x halving (``crop.py:44-47``)
and the detect-time ``/255`` + nearest resize fused in.  Detections come back in slide coordinates.

No CPU fallback: the product path needs the HIP library and a GPU."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .utils import non_max_suppression


class RegionTileStream:
    """Iterates over the tile rows of ``raster`` and yields ``(tiles [n,3,S,S] float32 on the device, [(ty, tx), ...])``.

    ``tile`` is the tile side on the (halved, if ``shrink`` == 2) slide, ``img_size`` the network input side."""

    def __init__(self, raster, tile=1536, img_size=1024, shrink=1):
        if not torch.cuda.is_available():
            raise _lib.AyError("no HIP device: RegionTileStream has no CPU fallback")
        assert raster.ndim == 3 and raster.shape[2] == 3 and raster.dtype == np.uint8, "uint8 [H,W,3] raster"
        assert shrink in (1, 2)
        self.raster, self.tile, self.S, self.shrink = raster, int(tile), int(img_size), int(shrink)
        H, W = raster.shape[0] // shrink, raster.shape[1] // shrink
        self.tiles_y, self.tiles_x = -(-H // self.tile), -(-W // self.tile)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        rows = self.tile * shrink
        self._pinned = [torch.empty(rows, raster.shape[1], 3, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._strips = [torch.empty(rows, raster.shape[1], 3, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self._copy = torch.cuda.Stream(device=self.dev)
        self._pool = ThreadPoolExecutor(max_workers=4)      # row chunks of one staging copy
        self._stager = ThreadPoolExecutor(max_workers=1)    # one strip ahead of the consumer
        self._chunk = -(-rows // 4)
        self._uploaded = [None, None]   # event: strip landed in _strips[k]
        self._consumed = [None, None]   # event: the ingest kernel that read _strips[k] is done

    def __len__(self):
        return self.tiles_y

    def _stage(self, j):
        """host side of strip j: raster rows -> pinned buffer (runs on the staging thread, under the consumer's GPU work)"""
        k = j & 1
        rows = self.tile * self.shrink
        src = self.raster[j * rows:(j + 1) * rows]
        n = src.shape[0]
        if self._uploaded[k] is not None:
            self._uploaded[k].synchronize()      # the host buffer is free again once its last copy has run
        # staging copy by NumPy (memcpy speed; torch's uint8 copy_ ran at a quarter of it), rows split over a few threads --
        # the copy releases the GIL -- so that one strip stages faster than the GPU consumes it
        dst = self._pinned[k].numpy()
        parts = [(a, min(a + self._chunk, n)) for a in range(0, n, self._chunk)]
        list(self._pool.map(lambda ab: np.copyto(dst[ab[0]:ab[1]], src[ab[0]:ab[1]]), parts))
        return n

    def _copy_to_device(self, j, n):
        k = j & 1
        if self._consumed[k] is not None:
            self._copy.wait_event(self._consumed[k])
        with torch.cuda.stream(self._copy):
            self._strips[k][:n].copy_(self._pinned[k][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        self._uploaded[k] = ev

    def __iter__(self):
        L = _lib.lib()
        main = torch.cuda.current_stream()
        dev_index = self.dev.index

        def stage(j):
            torch.cuda.set_device(dev_index)
            return self._stage(j)

        fut = self._stager.submit(stage, 0) if self.tiles_y else None
        for j in range(self.tiles_y):
            k = j & 1
            valid = fut.result()
            self._copy_to_device(j, valid)
            if j + 1 < self.tiles_y:  # staged while the consumer's model + NMS of this strip run
                fut = self._stager.submit(stage, j + 1)
            main.wait_event(self._uploaded[k])
            out = torch.empty(self.tiles_x, 3, self.S, self.S, device=self.dev, dtype=torch.float32)
            check(L.ay_ingest_region_tiles_u8(ptr(self._strips[k]), valid, self.raster.shape[1], self.raster.shape[1] * 3, self.shrink,
                                              self.tile, 1, self.tiles_x, self.S, ptr(out), _lib.stream_ptr()),
                  "ay_ingest_region_tiles_u8")
            done = torch.cuda.Event()
            done.record(main)
            self._consumed[k] = done
            yield out, [(j, i) for i in range(self.tiles_x)]


def detect_region(model, raster, tile=1536, img_size=1024, shrink=1, conf_thres=0.8, nms_thres=0.4, batch_size=64):
    """Detection over a whole raster: the loop of ``detect.py:88-105`` fed by :class:`RegionTileStream`.

    Returns a list of ``(ty, tx, boxes)`` with ``boxes [n,7]`` = (x1, y1, x2, y2, conf, cls_conf, cls_pred) in pixels of the
    (halved) slide -- the tile-local boxes of ``non_max_suppression`` scaled from the network size back to the tile
    (``rescale_boxes`` of a square tile is a pure scale) and shifted by the tile origin -- tiles without detections omitted."""
    results = []
    scale = float(tile) / float(img_size)
    model.eval()
    for tiles, coords in RegionTileStream(raster, tile, img_size, shrink):
        for s in range(0, tiles.shape[0], batch_size):
            with torch.no_grad():  # rows stay on the device; only the detections come back
                det = non_max_suppression(model.forward_device(tiles[s:s + batch_size]), conf_thres, nms_thres)
            for (ty, tx), d in zip(coords[s:s + batch_size], det):
                if d is None:
                    continue
                d = d.cpu()
                d[:, :4] *= scale
                d[:, [0, 2]] += tx * tile
                d[:, [1, 3]] += ty * tile
                results.append((ty, tx, d))
    return results
