"""multigriddet.models mirror (reference multigriddet/models/__init__.py)."""
from .multigriddet_darknet import (build_multigriddet_darknet, build_multigriddet_darknet_train, MultiGridDetModel,
                                   MultiGridDetTrainModel, Optimizer)


def list_available_models():
    return ["multigriddet_darknet"]


def create_model(name="multigriddet_darknet", **kwargs):
    if name not in ("multigriddet_darknet", "multigriddet_resnet"):     # "resnet" is the same graph
        raise ValueError(f"Unknown model '{name}'. Available: {list_available_models()}")
    return build_multigriddet_darknet(**kwargs)


__all__ = ["build_multigriddet_darknet", "build_multigriddet_darknet_train", "create_model", "list_available_models",
           "MultiGridDetModel", "MultiGridDetTrainModel", "Optimizer"]
