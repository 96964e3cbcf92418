"""`multigriddet` import-path shim: the reference's package name resolves to the MI355X-native implementation, so
`from multigriddet.losses import MultiGridLoss` etc. keep working unchanged (drop-in boundary)."""
import importlib
import sys

import multigriddet_amd as _impl

__version__ = _impl.__version__
for _sub in ("models", "losses", "data", "postprocess", "config", "trainers", "utils", "inference", "evaluation"):
    _m = importlib.import_module(f"multigriddet_amd.{_sub}")
    sys.modules[f"{__name__}.{_sub}"] = _m
    globals()[_sub] = _m
from multigriddet_amd.models import create_model, list_available_models  # noqa: E402,F401
