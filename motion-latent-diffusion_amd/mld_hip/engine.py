"""Per-device engine registry: HipMldDenoiser, HipMldVae, HipDDIMScheduler and the datamodule stub that are
instantiated separately from YAML (as the reference does, mld.py:56-83) all talk to ONE ``libmldhip``
handle per device, because the fused ``sample()`` needs every weight group in one place."""
from __future__ import annotations

from typing import Dict, Optional

from . import _lib

_engines: Dict[object, "_lib.Engine"] = {}
_defaults = dict(max_batch=64, max_frames=196)


def configure(**cfg):
    """Set engine capacity / scheduler fields used for engines created later (e.g. max_batch=128)."""
    _defaults.update(cfg)


def device_index(device) -> int:
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"mld_hip runs on MI355X only (got device '{d}'); there is no CPU path. "
                           "Move the module with .to('cuda').")
    return d.index if d.index is not None else torch.cuda.current_device()


def get_engine(device, **cfg) -> "_lib.Engine":
    key = device if isinstance(device, str) and device.startswith("inject:") else device_index(device)
    if key not in _engines:
        _engines[key] = _lib.Engine(device=key if isinstance(key, int) else 0, **{**_defaults, **cfg})
        _engines[key]._dirty = True
    return _engines[key]


def inject_engine(engine, name: str = "inject:test") -> str:
    """Test hook: register an externally built engine (the CPU suite passes the functional simulator's)."""
    _engines[name] = engine
    engine._dirty = True
    return name


def drop_engines():
    for e in _engines.values():
        e.close()
    _engines.clear()


def finalize_if_dirty(engine, stream: int = 0):
    if getattr(engine, "_dirty", True):
        engine.finalize(stream)
        engine._dirty = False


def current_stream_handle(tensor) -> int:
    import torch
    if tensor is not None and getattr(tensor, "is_cuda", False):
        return torch.cuda.current_stream(tensor.device).cuda_stream
    return 0
