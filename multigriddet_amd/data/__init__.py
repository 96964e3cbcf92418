"""multigriddet.data mirror (reference multigriddet/data/__init__.py)."""
from .generators import (MultiGridDataGenerator, preprocess_true_boxes, tf_preprocess_true_boxes,
                         load_annotation_lines, get_multiscale_list)

__all__ = ["MultiGridDataGenerator", "preprocess_true_boxes", "tf_preprocess_true_boxes", "load_annotation_lines",
           "get_multiscale_list"]
