"""YOLOv3 reward model: module tree with the reference's state-dict layout + the HIP/MFMA engine."""
from .model import DetectionModel, Model, yolov3  # noqa: F401
from .engine import YoloEngine  # noqa: F401
from .train_engine import YoloTrainEngine, YoloTrainPairEngine  # noqa: F401
