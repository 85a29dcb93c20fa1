"""Shared base of the drop-in nn.Modules: a parameter tree under the reference's state_dict names whose
arithmetic lives in libmldhip.  Parameters stay ordinary ``nn.Parameter``s so ``state_dict()`` /
``load_state_dict(strict=True)`` / ``.to(device)`` behave exactly like the reference modules'; before the
first use (and after any change) they are copied once, device-to-device, into the engine's weight arena."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import engine as _engine


class _Node(nn.Module):
    """Anonymous container so dotted checkpoint names map onto nested modules."""


class HipModule(nn.Module):
    _prefix = ""            # "denoiser." / "vae."  (Lightning checkpoint prefixes, base.py:96-127)
    _variant = "text"       # engine registry variant (mld_hip.engine)

    def __init__(self):
        super().__init__()
        self._arch: Dict[str, object] = {}           # mldhip_config fields this module's weights imply
        self._shared_arch: Dict[str, object] = {}    # fields of the other parts of the same model (set by MLD)
        self._engine_key: Optional[str] = None      # set by tests to an injected (simulator) engine
        self._synced_sig = None

    # ------------------------------------------------------------------ parameter tree
    def _register_tree(self, tensors: Dict[str, np.ndarray]):
        for name, value in tensors.items():
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Node())
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(torch.from_numpy(np.ascontiguousarray(value)).clone(),
                                                           requires_grad=False))

    # ------------------------------------------------------------------ engine plumbing
    def use_engine(self, key: str):
        """Bind to an injected engine (tests / simulator) instead of the per-device one."""
        self._engine_key = key
        self._synced_sig = None
        return self

    @property
    def engine(self):
        if self._engine_key is not None:
            return _engine.get_engine(self._engine_key)
        return _engine.get_engine(next(self.parameters()).device, self._variant, want={**self._shared_arch, **self._arch})

    def _set_arch(self, variant: str, **fields):
        self._variant = variant
        self._arch = dict(fields)

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def sync_weights(self):
        """Upload parameters to the engine if they changed since the last upload; finalize lazily."""
        eng = self.engine
        _engine.check_arch(eng, type(self).__name__, **self._arch)
        sig = (id(eng), self._signature())
        # One engine serves every module of the same (device, variant, architecture): a second model with the same
        # hyper-parameters overwrites this module's tensors in the shared arena, so "unchanged since MY last upload" is not
        # enough -- the engine remembers who uploaded each prefix last.
        owners = eng.__dict__.setdefault("_owner", {})
        if sig != self._synced_sig or owners.get(self._prefix) != (id(self), sig):
            for name, p in self.named_parameters():
                eng.load_tensor(self._prefix + name, p.data)
            eng._dirty = True
            self._synced_sig = sig
            owners[self._prefix] = (id(self), sig)
        _engine.finalize_if_dirty(eng, self._stream())
        return eng

    def _stream(self) -> int:
        p = next(self.parameters())
        return _engine.current_stream_handle(p)

    @staticmethod
    def _check(t: torch.Tensor, name: str) -> torch.Tensor:
        if t.dtype != torch.float32:
            raise TypeError(f"{name}: float32 expected (the reference samples in fp32, mld.py:297-307), got {t.dtype}")
        return t.contiguous()
