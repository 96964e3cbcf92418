"""multigriddet.trainers mirror (reference multigriddet/trainers/__init__.py)."""
from .trainer import MultiGridTrainer, CosineAnnealingWithWarmup

__all__ = ["MultiGridTrainer", "CosineAnnealingWithWarmup"]
