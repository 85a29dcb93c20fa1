"""Reading the reference's Lightning checkpoints without Lightning.

The released files (``1222_mld_humanml3d_FID041.ckpt`` ...) are ``torch.save``d dicts whose ``"state_dict"`` entry holds
the ``denoiser.* / vae.* / t2m_*`` tensors (mld/models/modeltype/base.py:96-127 strips ``text_encoder.*`` on save), next to
``hyper_parameters`` / ``callbacks`` / optimizer state that pickle classes of ``omegaconf`` and ``pytorch_lightning``.  None
of those packages is needed to sample, so this loader unpickles the file with a tolerant class resolver: a global that
cannot be imported becomes an inert placeholder instead of an ImportError, and only ``state_dict`` is returned.
(Like any pickle, a checkpoint can run code on load: read only files you trust.)
"""
from __future__ import annotations

import pickle
from typing import Dict

import torch


class _Placeholder:
    """Stands in for an object whose class is not importable here (omegaconf.DictConfig, Lightning callbacks, ...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):          # some pickles store factory functions (e.g. collections builders)
        return _Placeholder()

    def __reduce_ex__(self, protocol):
        return (_Placeholder, ())

    def append(self, item):               # list / dict subclasses are rebuilt through append / __setitem__ / extend
        pass

    def extend(self, items):
        pass

    def __setitem__(self, k, v):
        pass


def _placeholder_class(module: str, name: str):
    return type(name, (_Placeholder,), {"__module__": module})


class _TolerantUnpickler(pickle.Unpickler):
    missing = None

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            if _TolerantUnpickler.missing is not None:
                _TolerantUnpickler.missing.add(f"{module}.{name}")
            return _placeholder_class(module, name)


class _TolerantPickle:
    """The slice of the ``pickle`` module interface torch.load uses (``pickle_module=``)."""
    __name__ = "mld_hip_tolerant_pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def load_lightning_state_dict(path: str, report_missing: bool = False):
    """``ckpt["state_dict"]`` of a Lightning checkpoint (or the file itself when it already is a flat state dict), on CPU.
    With report_missing=True also returns the sorted list of pickled globals that were not importable."""
    missing = set()
    _TolerantUnpickler.missing = missing
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)
    finally:
        _TolerantUnpickler.missing = None
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    if not isinstance(sd, dict) or not all(isinstance(k, str) for k in sd):
        raise ValueError(f"{path}: no state_dict found (top-level keys: {list(ckpt)[:8] if isinstance(ckpt, dict) else type(ckpt)})")
    out: Dict[str, torch.Tensor] = {k: v for k, v in sd.items() if torch.is_tensor(v)}
    return (out, sorted(missing)) if report_missing else out
