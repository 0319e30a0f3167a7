"""``evaluate`` with the reference's signature and return convention (reference ``test.py:24-66``) + its CLI."""
import argparse

import numpy as np
import torch
from torch.utils.data import DataLoader

from .datasets import ListDataset
from .models import Darknet
from .parse_config import parse_data_config
from .stats import ap_per_class, get_batch_statistics
from .utils import load_classes, non_max_suppression, xywh2xyxy


def evaluate(model, path, iou_thres, conf_thres, nms_thres, img_size, batch_size):
    model.eval()
    dataset = ListDataset(path, img_size=img_size, multiscale=False)
    loader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=1, collate_fn=dataset.collate_fn)
    labels, sample_metrics = [], []
    for _, imgs, targets in loader:
        if targets is None:
            continue
        labels += targets[:, 1].tolist()
        targets[:, 2:] = xywh2xyxy(targets[:, 2:])
        targets[:, 2:] *= img_size
        with torch.no_grad():
            outputs = model(imgs)
            outputs = non_max_suppression(outputs, conf_thres=conf_thres, nms_thres=nms_thres)
        sample_metrics += get_batch_statistics(outputs, targets, iou_threshold=iou_thres)
    if len(sample_metrics) == 0:
        return None
    tp, scores, pred_labels = [np.concatenate(x, 0) for x in list(zip(*sample_metrics))]
    return ap_per_class(tp, scores, pred_labels, labels)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--model_def", type=str, default="config/yolov3.cfg")
    ap.add_argument("--data_config", type=str, default="config/coco.data")
    ap.add_argument("--weights_path", type=str, default="weights/yolov3.weights")
    ap.add_argument("--class_path", type=str, default="data/coco.names")
    ap.add_argument("--iou_thres", type=float, default=0.5)
    ap.add_argument("--conf_thres", type=float, default=0.5)
    ap.add_argument("--nms_thres", type=float, default=0.5)
    ap.add_argument("--n_cpu", type=int, default=8)
    ap.add_argument("--img_size", type=int, default=416)
    opt = ap.parse_args(argv)
    data_config = parse_data_config(opt.data_config)
    class_names = load_classes(data_config["names"])
    model = Darknet(opt.model_def).to("cuda")
    if opt.weights_path.endswith(".weights"):
        model.load_darknet_weights(opt.weights_path)
    else:
        model.load_state_dict(torch.load(opt.weights_path))
    res = evaluate(model, data_config["valid"], opt.iou_thres, opt.conf_thres, opt.nms_thres, opt.img_size, opt.batch_size)
    if res is None:
        print("no detections")
        return
    precision, recall, AP, f1, ap_class = res
    for i, c in enumerate(ap_class):
        print(f"+ Class '{c}' ({class_names[c]}) - AP: {AP[i]}")
    print(f"mAP: {AP.mean()}")


if __name__ == "__main__":
    main()
