"""Config / plugin system of the sampling path -- a small re-implementation of ``mld/config.py``.

The reference builds its config with OmegaConf (not installed here): ``base.yaml`` <- experiment yaml
<- every ``configs/<cfg.model.target>/*.yaml`` merged INTO ``cfg.model`` <- assets yaml, with
``${a.b}`` interpolations resolved against the root (mld/config.py:7-13,156-164), and instantiates
network parts from ``{target, params}`` nodes (mld/config.py:16-31).  This module reproduces exactly
that merge order and interpolation behaviour for the keys the hot path consumes, on plain pyyaml.
"""
from __future__ import annotations

import copy
import importlib
import os
import re
from typing import Any, Dict, Optional

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
_INTERP = re.compile(r"^\$\{([^}]+)\}$")


class Cfg(dict):
    """dict with attribute access (the OmegaConf DictConfig surface the reference code uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):  # noqa: D401  (same signature as DictConfig.get)
        return self[k] if k in self else default


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def merge(a: Dict[str, Any], b: Dict[str, Any]) -> Dict[str, Any]:
    """OmegaConf.merge semantics for dict trees: recursive for dicts, ``b`` wins otherwise (lists replaced)."""
    out = copy.deepcopy(a)
    for k, v in (b or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def _select(root, dotted):
    node = root
    for part in dotted.split("."):
        node = node[part]
    return node


_ROOT = object()


def resolve(root, node=_ROOT, _depth=0):
    """Resolve whole-value ``${a.b.c}`` interpolations (the only form the reference's YAMLs use)."""
    if _depth > 16:
        raise ValueError("interpolation cycle")
    node = root if node is _ROOT else node
    if isinstance(node, dict):
        return {k: resolve(root, v, _depth) for k, v in node.items()}
    if isinstance(node, list):
        return [resolve(root, v, _depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node.strip())
        if m:
            return resolve(root, copy.deepcopy(_select(root, m.group(1))), _depth + 1)
    return node


def _load_yaml(path):
    with open(path, "r") as f:
        return yaml.safe_load(f) or {}


def get_module_config(cfg_model: Dict[str, Any], target: str, config_dir: str) -> Dict[str, Any]:
    """mld/config.py:7-13 -- merge every yaml of configs/<target>/ into cfg.model (sorted for determinism)."""
    d = os.path.join(config_dir, target)
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".yaml"):
            cfg_model = merge(cfg_model, _load_yaml(os.path.join(d, fn)))
    return cfg_model


def load_config(cfg: Optional[str] = None, cfg_assets: Optional[str] = None, config_dir: str = CONFIG_DIR,
                overrides: Optional[Dict[str, Any]] = None) -> Cfg:
    """parse_args()'s file handling (mld/config.py:156-164) without argparse.

    ``cfg`` / ``cfg_assets`` default to config_mld_humanml3d.yaml / assets.yaml of ``config_dir``.
    ``overrides`` is merged last (dotted keys allowed: {"model.guidance_scale": 5.0}).
    """
    cfg = cfg or os.path.join(config_dir, "config_mld_humanml3d.yaml")
    cfg_assets = cfg_assets or os.path.join(config_dir, "assets.yaml")
    base = _load_yaml(os.path.join(config_dir, "base.yaml"))
    exp = merge(base, _load_yaml(cfg))
    model = get_module_config(exp["model"], exp["model"].get("target", "modules"), config_dir)
    out = merge(merge(exp, {"model": model}), _load_yaml(cfg_assets))
    for k, v in (overrides or {}).items():
        node = out
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return _wrap(resolve(out))


def get_obj_from_str(string: str):
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    """mld/config.py:24-31 -- ``{target: dotted.Class, params: {...}}`` -> ``Class(**params)``."""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**dict(config.get("params", dict()) or {}))
