"""Generator for the YOLOv3 (Darknet-53 + FPN) cfg the paper trains.

The reference ships the topology as a text file (``config/yolov3-custom.cfg``)
produced by ``config/create_custom_model.sh`` for a given class count.  This
module emits the same block sequence for any ``num_classes`` so tests and the
bench need no file from the reference tree (SURVEY.md App. A gives the layer
map this follows).
"""
import os

ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"


def _conv(filters, size, stride=1, bn=True, act="leaky"):
    lines = ["[convolutional]"]
    if bn:
        lines.append("batch_normalize=1")
    lines += [f"filters={filters}", f"size={size}", f"stride={stride}", "pad=1", f"activation={act}", ""]
    return lines


def _residual_stage(channels, n_blocks):
    out = _conv(channels, 3, stride=2)
    for _ in range(n_blocks):
        out += _conv(channels // 2, 1)
        out += _conv(channels, 3)
        out += ["[shortcut]", "from=-3", "activation=linear", ""]
    return out


def _yolo(mask, num_classes):
    return [
        "[yolo]", f"mask = {mask}", f"anchors = {ANCHORS}", f"classes={num_classes}", "num=9",
        "jitter=.3", "ignore_thresh = .7", "truth_thresh = 1", "random=1", "",
    ]


def yolov3_cfg_text(num_classes=2, size=416):
    head = 3 * (5 + num_classes)
    L = [
        "[net]", "batch=16", "subdivisions=1", f"width={size}", f"height={size}", "channels=3",
        "momentum=0.9", "decay=0.0005", "angle=0", "saturation = 1.5", "exposure = 1.5", "hue=.1",
        "learning_rate=0.001", "burn_in=1000", "max_batches = 500200", "policy=steps",
        "steps=400000,450000", "scales=.1,.1", "",
    ]
    L += _conv(32, 3)
    for ch, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        L += _residual_stage(ch, n)
    # neck + three detection heads (stride 32, 16, 8)
    for scale_i, (ch, mask, route_from) in enumerate(((512, "6,7,8", None), (256, "3,4,5", 61), (128, "0,1,2", 36))):
        if route_from is not None:
            L += ["[route]", "layers = -4", ""]
            L += _conv(ch, 1)
            L += ["[upsample]", "stride=2", ""]
            L += ["[route]", f"layers = -1, {route_from}", ""]
        for _ in range(3):
            L += _conv(ch, 1)
            L += _conv(ch * 2, 3)
        L += _conv(head, 1, bn=False, act="linear")
        L += _yolo(mask, num_classes)
    return "\n".join(L) + "\n"


def write_cfg(num_classes=2, directory=None, size=416):
    """Write (once) and return the path of the generated cfg."""
    directory = directory or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_generated")
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"yolov3-custom-c{num_classes}.cfg")
    text = yolov3_cfg_text(num_classes, size)
    if not os.path.exists(path) or open(path).read() != text:
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "w") as fh:
            fh.write(text)
        os.replace(tmp, path)
    return path
