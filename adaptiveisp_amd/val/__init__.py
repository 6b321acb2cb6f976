"""Eval harness of the hot path: NMS, box geometry, mAP, and the evaluation loop of
`yolov3/val_adaptiveisp.py:run` with the ISP + detector on the HIP path."""
from .boxes import clip_boxes, letterbox_geometry, letterbox_pad, scale_boxes, xywh2xyxy, xyxy2xywh  # noqa: F401
from .metrics import ap_per_class, box_iou, compute_ap, process_batch, smooth  # noqa: F401
from .nms import hip_nms, non_max_suppression  # noqa: F401
from .harness import run_eval  # noqa: F401
from .loader import LODImages, letterbox, load_image, resize_area_u8, resize_linear_u8  # noqa: F401
