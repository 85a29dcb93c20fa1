"""Minimal stand-in for ``HumanML3DDataModule`` (mld/data/HumanML3D.py:11-75): the sampling path only needs
``nfeats``, ``njoints``, the normalisation vectors and ``feats2joints`` -- and the reference's datamodule
cannot even be constructed without the full dataset, GloVe and pytorch_lightning."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import engine as _engine
from . import synthetic as syn


class HipDataModule:
    name = "humanml3d"

    def __init__(self, cfg=None, mean: Optional[np.ndarray] = None, std: Optional[np.ndarray] = None,
                 nfeats: int = 263, njoints: int = 22, engine_key: Optional[str] = None, name: str = "humanml3d",
                 nclasses: int = 12, variant: Optional[str] = None):
        """name 'humanml3d' (263-d features, 22 joints) or 'humanact12' (rot6d 25x6 = 150-d features, 12 classes;
        mld/data/HumanAct12.py) -- the latter only carries shapes: its feats2joints needs SMPL."""
        self.name, self.nclasses = name, nclasses
        self.variant = variant          # engine registry variant; filled in by MLD when left None
        self.nfeats, self.njoints = nfeats, njoints
        if mean is None or std is None:
            root = None
            try:
                root = cfg.DATASET.HUMANML3D.ROOT
            except Exception:
                pass
            if root and os.path.exists(os.path.join(root, "Mean.npy")):      # get_data.py:38-40
                mean = np.load(os.path.join(root, "Mean.npy"))
                std = np.load(os.path.join(root, "Std.npy"))
                self.stats = "dataset"
            else:
                mean, std = syn.make_mean_std(nfeats)
                self.stats = "synthetic"
        self.mean = np.asarray(mean, np.float32)
        self.std = np.asarray(std, np.float32)
        self.hparams = type("H", (), {"mean": self.mean, "std": self.std})()
        self._engine_key = engine_key
        self._shared_arch = {}          # architecture fields of the model this datamodule serves (set by MLD)
        self._loaded_on = None

    def _engine(self, device):
        eng = _engine.get_engine(self._engine_key if self._engine_key is not None else device, self.variant or "text",
                                 want=self._shared_arch)
        owners = eng.__dict__.setdefault("_owner", {})           # see HipModule.sync_weights: engines are shared per architecture
        if self._loaded_on is not eng or owners.get("mean/std") != id(self):
            eng.load_tensor("mean", self.mean)
            eng.load_tensor("std", self.std)
            eng._dirty = True
            self._loaded_on = eng
            owners["mean/std"] = id(self)
        return eng

    def feats2joints(self, features: torch.Tensor, mask=None) -> torch.Tensor:
        """[B, T, nfeats] -> [B, T, njoints, 3] (HumanML3D.py:41-45 + recover_from_ric), on the tensor's device."""
        if self.name != "humanml3d":
            raise NotImplementedError(f"feats2joints of '{self.name}' maps rot6d features through the SMPL body model "
                                      "(mld/transforms/rots2joints/smplh.py); SMPL is an external asset and out of scope")
        if features.dtype != torch.float32:
            raise TypeError("feats2joints expects float32 features (recover_from_ric's index_put needs fp32)")
        f = features.contiguous()
        eng = self._engine(f.device)
        _engine.finalize_if_dirty(eng, _engine.current_stream_handle(f))
        out = torch.empty(*f.shape[:2], self.njoints, 3, dtype=torch.float32, device=f.device)
        eng.feats2joints(f, f.shape[0], f.shape[1], out, _engine.current_stream_handle(f))
        return out
