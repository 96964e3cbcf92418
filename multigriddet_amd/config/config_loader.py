"""YAML configuration loading with the reference's semantics (reference
multigriddet/config/config_loader.py:14-114): yaml.safe_load, deep merge with override priority,
required-key validation per config type, relative-path resolution for .yaml/.txt/.h5 values."""
import copy
import os
from pathlib import Path
from typing import Any, Dict

import yaml

_REQUIRED = {"training": ["model_config", "data", "training"], "inference": ["model_config", "input", "detection"],
             "evaluation": ["model_config", "data", "evaluation"]}


class ConfigLoader:
    @staticmethod
    def load_config(config_path: str) -> Dict[str, Any]:
        if not os.path.exists(config_path):
            raise FileNotFoundError(f"Config file not found: {config_path}")
        with open(config_path, "r") as f:
            return yaml.safe_load(f)

    load_model_config = load_config

    @staticmethod
    def merge_configs(base_config: Dict, override_config: Dict) -> Dict:
        out = copy.deepcopy(base_config)
        for k, v in override_config.items():
            out[k] = ConfigLoader.merge_configs(out[k], v) if isinstance(out.get(k), dict) and isinstance(v, dict) else v
        return out

    @staticmethod
    def validate_config(config: Dict[str, Any], config_type: str = "general") -> bool:
        required = list(_REQUIRED.get(config_type, []))
        if config_type == "model":
            required = ["model"]
            kind = config.get("model", {}).get("type")
            if kind in ("preset", "custom"):
                required.append(kind)       # reference checks these at top level (config_loader.py:57-63)
        if config_type == "training" and "loss_option" in config.get("training", {}):
            lo = config["training"]["loss_option"]
            if lo not in [1, 2, 3]:
                raise ValueError(f"Invalid loss_option: {lo}. Must be 1, 2, or 3.")
        for k in required:
            if k not in config:
                raise KeyError(f"Missing required key '{k}' in {config_type} config")
        return True

    @staticmethod
    def resolve_paths(config: Dict[str, Any], base_dir: str = ".") -> Dict[str, Any]:
        base = Path(base_dir).resolve()

        def res(v):
            if isinstance(v, str) and v.endswith((".yaml", ".txt", ".h5")) and not os.path.isabs(v):
                return str(base / v)
            if isinstance(v, dict):
                return {k: res(x) for k, x in v.items()}
            if isinstance(v, list):
                return [res(x) for x in v]
            return v

        return res(config)

    @staticmethod
    def load_and_validate(config_path: str, config_type: str = "general") -> Dict[str, Any]:
        cfg = ConfigLoader.load_config(config_path)
        cfg = ConfigLoader.resolve_paths(cfg, os.path.dirname(config_path))
        ConfigLoader.validate_config(cfg, config_type)
        return cfg
