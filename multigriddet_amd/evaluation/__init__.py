"""Evaluation: MultiGridEvaluator and the mAP metrics (reference multigriddet/evaluation/__init__.py)."""
from .evaluator import MultiGridEvaluator
from .metrics import calculate_map, calculate_iou_matrix, print_map_results

__all__ = ["MultiGridEvaluator", "calculate_map", "calculate_iou_matrix", "print_map_results"]
