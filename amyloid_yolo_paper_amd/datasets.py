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


def label_path_of(image_path):
    """the annotation contract: .../images/x.jpg|png -> .../labels/x.txt"""
    stem = image_path.replace("images", "labels")
    for ext in (".png", ".jpg"):
        stem = stem.replace(ext, ".txt")
    return stem


def read_sample(image_path):
    """(uint8 HWC image, boxes [n,5] = class cx cy w h) of one tile, or None when either file cannot be read (such samples
    are skipped by the collate step, as in the reference: ``utils/datasets.py:93-95,107-109``)"""
    try:
        img = np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8)
    except Exception as exc:
        warnings.warn(f"skipping {image_path}: image not readable ({exc})")
        return None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # an empty label file is a tile without boxes
            boxes = np.loadtxt(label_path_of(image_path)).reshape(-1, 5)
    except Exception as exc:
        warnings.warn(f"skipping {image_path}: labels not readable ({exc})")
        return None
    return img, boxes


class SizeSchedule:
    """multi-scale training (reference ``utils/datasets.py:78-79,131-133``): every 10th batch a new side is drawn from
    [base - 96, base + 96] in steps of 32"""

    def __init__(self, base, enabled):
        self.size = base
        self.enabled = enabled
        self.choices = list(range(base - 96, base + 96 + 1, 32))
        self.batches = 0

    def next(self):
        self.batches += 1
        if self.enabled and self.batches % 10 == 0:
            self.size = random.choice(self.choices)
        return self.size


class ListDataset(Dataset):
    """Training / evaluation set given as a text file of image paths (reference ``utils/datasets.py:65-143``); items are
    (path, float image [3,S',S'] padded to square, targets [n,6] with column 0 left for the sample index)."""

    def __init__(self, list_path, img_size=416, multiscale=True):
        with open(list_path) as fh:
            self.img_files = [line.strip() for line in fh if line.strip()]
        self.schedule = SizeSchedule(img_size, multiscale)

    @property
    def img_size(self):
        return self.schedule.size

    def __len__(self):
        return len(self.img_files)

    def __getitem__(self, index):
        path = self.img_files[index % len(self.img_files)]
        sample = read_sample(path)
        if sample is None:
            return None
        img, targets = default_transform(*sample)
        return path, img, targets

    def collate_fn(self, batch):
        """drop unreadable samples, bring every image to the batch's side (nearest), number the targets by sample"""
        size = self.schedule.next()
        kept = [item for item in batch if item is not None]
        paths = tuple(item[0] for item in kept)
        imgs = torch.stack([resize(item[1], size) for item in kept])
        for k, item in enumerate(kept):
            item[2][:, 0] = k
        return paths, imgs, torch.cat([item[2] for item in kept], 0)
