"""Per-device engine registry: HipMldDenoiser, HipMldVae, HipDDIMScheduler and the datamodule stub that are
instantiated separately from YAML (as the reference does, mld.py:56-83) all talk to ONE ``libmldhip``
handle per (device, model variant), because the fused ``sample()`` needs every weight group in one place.

Engines are looked up by (device, variant) AND by the architecture fields the asking module needs (`want`): a second
model of the same variant with other hyper-parameters gets its own engine instead of silently changing -- or failing
against -- the first one's.  ``configure`` only sets defaults (capacity: max_batch / max_frames / max_in_flight ...) for
engines created later.

Variants: "text" = config_mld_humanml3d (MldDenoiser text condition + MldVae), "action" = config_mld_humanact12
(MldDenoiser action condition + ActorVae), "novae" = config_novae_humanml3d (trans_dec MldDenoiser on raw motion, DDPM).  Modules keep their own architecture fields (``HipModule._arch``; ``MLD`` merges those of its parts with the
scheduler / guidance fields into one shared dict) and pass them as `want`; ``check_arch`` re-verifies them before every use."""
from __future__ import annotations

from typing import Dict

from . import _lib

_engines: Dict[object, "_lib.Engine"] = {}
_defaults = {
    "text": dict(max_batch=64, max_frames=196),
    "action": dict(max_batch=64, max_frames=60, condition=_lib.COND_ACTION, vae_arch=_lib.VAE_ACTOR, num_layers=15,
                   vae_num_layers=6, nclasses=12, nfeats=150),
    "novae": dict(max_batch=64, max_frames=196, latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                  scheduler_type=_lib.SCHED_DDPM, num_inference_steps=1000, steps_offset=0),
}


def configure(variant: str = "text", **cfg):
    """Set engine capacity / scheduler / architecture fields for engines of `variant` created later."""
    _defaults[variant].update(cfg)


def check_arch(engine, who: str, **want):
    """Raise when a live engine was created with other architecture fields than module `who` needs."""
    bad = {k: (getattr(engine.cfg, k), v) for k, v in want.items()
           if (abs(getattr(engine.cfg, k) - v) > 1e-6 if isinstance(v, float) else getattr(engine.cfg, k) != v)}
    if bad:
        raise RuntimeError(f"{who}: engine/module mismatch " + ", ".join(f"{k}: engine {a} vs module {b}" for k, (a, b) in bad.items())
                           + "; construct the modules before the first use of the engine, or call mld_hip.engine.drop_engines()")


def device_index(device) -> int:
    import torch
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"mld_hip runs on MI355X only (got device '{d}'); there is no CPU path. "
                           "Move the module with .to('cuda').")
    return d.index if d.index is not None else torch.cuda.current_device()


def _matches(engine, want) -> bool:
    for k, v in want.items():
        have = getattr(engine.cfg, k)
        if (abs(have - v) > 1e-6 * max(1.0, abs(v))) if isinstance(v, float) else have != v:
            return False
    return True


def get_engine(device, variant: str = "text", want=None, **cfg) -> "_lib.Engine":
    """The engine of (device, variant) whose config has every field of `want`; created (defaults + want + cfg) if none does."""
    if isinstance(device, str) and device.startswith("inject:"):
        return _engines[device]
    dev = device_index(device)
    want = dict(want or {})
    mine = [k for k in _engines if isinstance(k, tuple) and k[:2] == (dev, variant)]
    for k in mine:
        if _matches(_engines[k], want):
            return _engines[k]
    key = (dev, variant, len(mine))
    _engines[key] = _lib.Engine(device=dev, **{**_defaults[variant], **want, **cfg})
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
