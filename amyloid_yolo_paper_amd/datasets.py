"""Host-side tile/label readers with the reference's conventions (reference ``utils/datasets.py``,
``utils/transforms.py:68-118``): PIL -> RGB uint8 -> /255 CHW float32 -> zero pad to square -> nearest resize;
labels ``class cx cy w h`` normalised, re-normalised for the padding; ``collate_fn`` writes the sample index into
column 0 and (optionally) re-draws the size every 10th batch.  The imgaug augmentation pipeline of the reference is
out of scope: this loader is the deterministic DEFAULT_TRANSFORMS path."""
import glob
import random
import warnings

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import Dataset


def pad_to_square(img, pad_value=0.0):
    """img [C,H,W] -> centre-padded square, returns (img, (left, right, top, bottom))"""
    _, h, w = img.shape
    diff = abs(h - w)
    p1, p2 = diff // 2, diff - diff // 2
    pad = (0, 0, p1, p2) if h <= w else (p1, p2, 0, 0)
    return F.pad(img, pad, "constant", value=pad_value), pad


def resize(image, size):
    return F.interpolate(image.unsqueeze(0), size=size, mode="nearest").squeeze(0)


def ingest_tiles_device(tiles_u8, img_size, pad_value=0.0, out=None):
    """uint8 HWC tiles [B,H,W,3] (NumPy or tensor, host or device) -> float32 [B,3,S,S] on the current HIP device:
    ``to_tensor`` + ``pad_to_square`` + ``resize`` of this module (the reference's DEFAULT_TRANSFORMS + Resize image path)
    in ONE device pass (``ay_ingest_tiles_u8``); the host uploads 3 bytes per pixel.  No CPU fallback."""
    from . import _lib
    from ._lib import check, ptr
    if not torch.cuda.is_available():
        raise _lib.AyError("no HIP device: ingest_tiles_device has no CPU fallback (use to_tensor/pad_to_square/resize on the host)")
    t = torch.as_tensor(tiles_u8)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    assert t.dtype == torch.uint8 and t.dim() == 4 and t.shape[-1] == 3, "uint8 [B,H,W,3] tiles"
    dev = torch.device("cuda", torch.cuda.current_device())
    t = t.to(dev, non_blocking=True).contiguous()
    B, H, W, _ = t.shape
    if out is None:
        out = torch.empty(B, 3, img_size, img_size, device=dev, dtype=torch.float32)
    assert out.shape == (B, 3, img_size, img_size) and out.is_contiguous() and out.dtype == torch.float32
    check(_lib.lib().ay_ingest_tiles_u8(ptr(t), B, H, W, img_size, float(pad_value), ptr(out), _lib.stream_ptr()), "ay_ingest_tiles_u8")
    return out


def to_tensor(img_u8):
    """HWC uint8 -> CHW float32 /255 (what torchvision's ToTensor does)"""
    return torch.from_numpy(np.ascontiguousarray(img_u8.transpose(2, 0, 1))).float().div(255.0)


def default_transform(img_u8, boxes):
    """AbsoluteLabels -> PadSquare -> RelativeLabels -> ToTensor of the reference: boxes [n,5] (class cx cy w h)."""
    h, w, _ = img_u8.shape
    img, pad = pad_to_square(to_tensor(img_u8))
    _, ph, pw = img.shape
    boxes = np.array(boxes, dtype=np.float64, copy=True).reshape(-1, 5)
    if len(boxes):
        x1 = w * (boxes[:, 1] - boxes[:, 3] / 2) + pad[0]
        y1 = h * (boxes[:, 2] - boxes[:, 4] / 2) + pad[2]
        x2 = w * (boxes[:, 1] + boxes[:, 3] / 2) + pad[0]
        y2 = h * (boxes[:, 2] + boxes[:, 4] / 2) + pad[2]
        boxes[:, 1] = ((x1 + x2) / 2) / pw
        boxes[:, 2] = ((y1 + y2) / 2) / ph
        boxes[:, 3] = (x2 - x1) / pw
        boxes[:, 4] = (y2 - y1) / ph
    targets = torch.zeros((len(boxes), 6))
    targets[:, 1:] = torch.from_numpy(boxes).float()
    return img, targets


class ImageFolder(Dataset):
    """sorted glob of a folder -> (path, tensor [3,S,S]) (reference ``utils/datasets.py:40-62`` + Resize)"""

    def __init__(self, folder_path, img_size=416, raw_u8=False):
        self.files = sorted(glob.glob("%s/*.*" % folder_path))
        self.img_size = img_size
        self.raw_u8 = raw_u8  # hand out the decoded uint8 HWC tile; ``ingest_tiles_device`` does the rest on the GPU

    def __getitem__(self, index):
        path = self.files[index % len(self.files)]
        img = np.array(Image.open(path).convert("RGB"), dtype=np.uint8)
        if self.raw_u8:
            return path, torch.from_numpy(img)
        img, _ = default_transform(img, np.zeros((0, 5)))
        return path, resize(img, self.img_size)

    def __len__(self):
        return len(self.files)


class ListDataset(Dataset):
    """list file of image paths; labels at images->labels, .jpg/.png->.txt (reference ``utils/datasets.py:65-143``)"""

    def __init__(self, list_path, img_size=416, multiscale=True):
        with open(list_path, "r") as fh:
            self.img_files = [l for l in fh.readlines() if l.strip()]
        self.label_files = [p.replace("images", "labels").replace(".png", ".txt").replace(".jpg", ".txt") for p in self.img_files]
        self.img_size = img_size
        self.multiscale = multiscale
        self.min_size = self.img_size - 3 * 32
        self.max_size = self.img_size + 3 * 32
        self.batch_count = 0

    def __getitem__(self, index):
        try:
            img_path = self.img_files[index % len(self.img_files)].rstrip()
            img = np.array(Image.open(img_path).convert("RGB"), dtype=np.uint8)
        except Exception:
            print(f"Could not read image '{img_path}'.")
            return None
        try:
            label_path = self.label_files[index % len(self.img_files)].rstrip()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                boxes = np.loadtxt(label_path).reshape(-1, 5)
        except Exception:
            print(f"Could not read label '{label_path}'.")
            return None
        img, targets = default_transform(img, boxes)
        return img_path, img, targets

    def collate_fn(self, batch):
        self.batch_count += 1
        batch = [d for d in batch if d is not None]
        paths, imgs, targets = list(zip(*batch))
        if self.multiscale and self.batch_count % 10 == 0:
            self.img_size = random.choice(range(self.min_size, self.max_size + 1, 32))
        imgs = torch.stack([resize(img, self.img_size) for img in imgs])
        for i, boxes in enumerate(targets):
            boxes[:, 0] = i
        return paths, imgs, torch.cat(targets, 0)

    def __len__(self):
        return len(self.img_files)
