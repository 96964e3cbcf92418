"""multigriddet.config mirror (reference multigriddet/config/__init__.py)."""
from .config_loader import ConfigLoader
from .model_builder import (build_model_for_inference, build_model_for_training, build_model_from_config,
                            create_optimizer_from_config, get_model_info)

__all__ = ["ConfigLoader", "build_model_for_inference", "build_model_for_training", "build_model_from_config",
           "create_optimizer_from_config", "get_model_info"]
