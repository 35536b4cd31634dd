"""Minimal `_target_` instantiation so the reference's frozen Hydra YAML
(e.g. pretrained/bunny_smoke/.hydra/config.yaml) works without hydra/omegaconf,
which are not installed here.  With hydra present, hydra.utils.instantiate
resolves the same `_target_` strings to the same classes."""
import importlib
from typing import Any, Mapping


def to_plain(cfg: Any) -> Any:
    """DictConfig / nested mappings -> plain python containers."""
    if isinstance(cfg, Mapping):
        return {k: to_plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)) or type(cfg).__name__ == "ListConfig":
        return [to_plain(v) for v in cfg]
    return cfg


def instantiate(cfg: Mapping, **kwargs: Any) -> Any:
    args = to_plain(cfg)
    args.update(kwargs)
    target = args.pop("_target_")
    args.pop("_recursive_", None)
    module, name = target.rsplit(".", 1)
    return getattr(importlib.import_module(module), name)(**args)
