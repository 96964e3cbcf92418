"""multigriddet.inference mirror (reference multigriddet/inference/__init__.py)."""
from .inference_engine import MultiGridInference

__all__ = ["MultiGridInference"]
