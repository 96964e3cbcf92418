"""multigriddet.postprocess mirror (reference multigriddet/postprocess/__init__.py)."""
from .multigrid_decode import MultiGridDecoder

__all__ = ["MultiGridDecoder"]
