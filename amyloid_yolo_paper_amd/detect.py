"""``detect()``: the body of the reference's ``detect.py`` script as a function with the same flags
(reference ``detect.py:30-105``; the matplotlib rendering and the paper's merge/CAA post-processing at ``:107-171`` are
out of scope).  Returns ``(image paths, list of [n,7] tensors | None)`` with boxes rescaled to each
image's original size."""
import argparse
import datetime
import time

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader

from .datasets import ImageFolder, ingest_tiles_device
from .models import Darknet
from .utils import load_classes, non_max_suppression, rescale_boxes


def detect(image_folder="data/samples", model_def="config/yolov3.cfg", weights_path="weights/yolov3.weights",
           class_path=None, conf_thres=0.8, nms_thres=0.4, batch_size=1, n_cpu=0, img_size=416, precision="bf16",
           rescale=True, verbose=True, device_ingest=True, merge_boxes=False, merge_on_device=False, write_CAA_detections_to_pickle=False,
           filter_CAA_detections_by_model=False):
    """``precision``: "bf16" | "fp16" (the two 16-bit MFMA storage types) | "fp32" (reference-precision parity path).
    ``device_ingest``: upload the decoded uint8 tiles and do /255 + pad-to-square + nearest resize on the GPU
    (``ay_ingest_tiles_u8``, bit-identical to the host transforms); batches of mixed image sizes are ingested one size at a time.
    ``merge_boxes``: the reference's ``--merge_boxes True`` (``detect.py:131-133``): union-merge overlapping same-class boxes
    after the rescale (``postprocess.merge_detections``: the reference's container semantics on the host; ``merge_on_device=True``
    sends all images through one ``ay_merge_detections`` launch instead -- same rows as sets, explicit row order).
    ``write_CAA_detections_to_pickle`` / ``filter_CAA_detections_by_model`` (reference ``detect.py:43-44,134-141``) belong to the
    paper's second-stage CAA classifier (``core.filterDetectionsByCAAModel`` / ``writeCAADetectionsToPickle``: cv2, skimage and
    a Git-LFS model pickle): accepted so that existing command lines parse, refused with a clear error when switched on."""
    for flag, on in (("write_CAA_detections_to_pickle", write_CAA_detections_to_pickle),
                     ("filter_CAA_detections_by_model", filter_CAA_detections_by_model)):
        if on:
            raise NotImplementedError(
                f"--{flag} True: the second-stage CAA model of the reference (core.py:425-480; needs cv2, skimage and the Git-LFS "
                "pickle CAA_consensus_of_2_model) is outside this library's scope (SURVEY.md section 2); run detect() without it "
                "and post-process the returned detections with the reference's core.py")
    model = Darknet(model_def, img_size=img_size, precision=precision).to("cuda")
    if weights_path.endswith(".weights"):
        model.load_darknet_weights(weights_path)
    else:
        model.load_state_dict(torch.load(weights_path))
    model.eval()
    loader = DataLoader(ImageFolder(image_folder, img_size=img_size, raw_u8=device_ingest), batch_size=batch_size, shuffle=False,
                        num_workers=n_cpu, collate_fn=(lambda b: tuple(zip(*b))) if device_ingest else None)
    classes = load_classes(class_path) if class_path else None
    paths, results = [], []
    prev = time.time()
    for batch_i, (img_paths, imgs) in enumerate(loader):
        if device_ingest:  # list of uint8 [H,W,3] tiles -> [B,3,S,S] on the device, grouped by source size
            out = torch.empty(len(imgs), 3, img_size, img_size, device="cuda", dtype=torch.float32)
            by_shape = {}
            for i, t in enumerate(imgs):
                by_shape.setdefault(tuple(t.shape), []).append(i)
            for idx in by_shape.values():
                out[idx] = ingest_tiles_device(torch.stack([imgs[i] for i in idx]), img_size)
            imgs = out
        with torch.no_grad():
            if device_ingest:
                # the [B,N,5+C] rows stay on the device: only the detections come back (the reference's `model(x)` returns the
                # full CPU tensor, 2 MB per 1024^2 tile, which nothing downstream of NMS reads)
                dets = [None if d is None else d.cpu() for d in non_max_suppression(model.forward_device(imgs), conf_thres, nms_thres)]
            else:
                dets = non_max_suppression(model(imgs), conf_thres, nms_thres)
        now = time.time()
        if verbose:
            print("\t+ Batch %d, Inference Time: %s" % (batch_i, datetime.timedelta(seconds=now - prev)))
        prev = now
        paths.extend(img_paths)
        results.extend(dets)
    if rescale:
        for path, det in zip(paths, results):
            if det is not None:
                w, h = Image.open(path).size
                rescale_boxes(det, img_size, (h, w))
    if merge_boxes:
        from .postprocess import merge_detections, merge_detections_batch_device
        if merge_on_device:
            results = merge_detections_batch_device(results)
        else:
            results = [None if det is None else merge_detections(det) for det in results]
    return paths, results, classes


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--image_folder", type=str, default="data/samples")
    ap.add_argument("--model_def", type=str, default="config/yolov3.cfg")
    ap.add_argument("--weights_path", type=str, default="weights/yolov3.weights")
    ap.add_argument("--class_path", type=str, default="data/coco.names")
    ap.add_argument("--conf_thres", type=float, default=0.8)
    ap.add_argument("--nms_thres", type=float, default=0.4)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--n_cpu", type=int, default=0)
    ap.add_argument("--img_size", type=int, default=416)
    ap.add_argument("--checkpoint_model", type=str)
    ap.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--merge_boxes", type=str, default="False", help="merge overlapping boxes of the same class (reference detect.py:42)")
    ap.add_argument("--write_CAA_detections_to_pickle", type=str, default="False", help="reference detect.py:43 -- out of scope: errors when True")
    ap.add_argument("--filter_CAA_detections_by_model", type=str, default="False", help="reference detect.py:44 -- out of scope: errors when True")
    opt = ap.parse_args(argv)
    paths, results, classes = detect(opt.image_folder, opt.model_def, opt.weights_path, opt.class_path, opt.conf_thres,
                                     opt.nms_thres, opt.batch_size, opt.n_cpu, opt.img_size, opt.precision,
                                     merge_boxes=opt.merge_boxes == "True",
                                     write_CAA_detections_to_pickle=opt.write_CAA_detections_to_pickle == "True",
                                     filter_CAA_detections_by_model=opt.filter_CAA_detections_by_model == "True")
    for path, det in zip(paths, results):
        print(f"Image: '{path}'")
        if det is None:
            continue
        for x1, y1, x2, y2, conf, cls_conf, cls_pred in det.tolist():
            name = classes[int(cls_pred)] if classes else int(cls_pred)
            print(f"\t+ Label: {name}, Conf: {cls_conf:.5f}, box: ({x1:.1f}, {y1:.1f}, {x2:.1f}, {y2:.1f})")


if __name__ == "__main__":
    main()
