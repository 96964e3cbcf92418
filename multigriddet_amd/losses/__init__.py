"""multigriddet.losses mirror (reference multigriddet/losses/__init__.py)."""
from .multigrid_loss import MultiGridLoss, multigriddet_loss

__all__ = ["MultiGridLoss", "multigriddet_loss"]
