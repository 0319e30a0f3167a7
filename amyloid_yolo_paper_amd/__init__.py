"""MI355X-native YOLOv3 hot path for amyloid-plaque tiles (drop-in for keiserlab/amyloid-yolo-paper's
``models.Darknet`` / ``utils.utils`` call surface).  See DESIGN.md."""
from . import _lib  # noqa: F401
from .models import Darknet  # noqa: F401
from .utils import (bbox_iou, bbox_iou_pairwise, bbox_wh_iou, non_max_suppression, rescale_boxes,  # noqa: F401
                    weights_init_normal, xywh2xyxy, load_classes, to_cpu)
