"""multigriddet.utils mirror: anchors/classes parsing and the inference letterbox."""
from .anchors import load_anchors, load_classes, compute_class_weights
from .preprocessing import preprocess_image, letterbox_resize

__all__ = ["load_anchors", "load_classes", "compute_class_weights", "preprocess_image", "letterbox_resize"]
