"""``HipMldVae`` -- drop-in for ``mld.models.architectures.mld_vae.MldVae`` (arch encoder_decoder, PE mld).

``decode(z, lengths)`` (mld_vae.py:186-248) runs on the HIP engine.  ``encode`` is the next row of the
scope table and raises until it lands; the encoder weights are still part of the
``state_dict`` so a released checkpoint loads with ``strict=True``.
"""
from __future__ import annotations

from typing import List

import torch

from . import synthetic as syn
from ._module import HipModule


class HipMldVae(HipModule):
    _prefix = "vae."

    def __init__(self, ablation, nfeats: int, latent_dim: list = [1, 256], ff_size: int = 1024, num_layers: int = 9,
                 num_heads: int = 4, dropout: float = 0.1, arch: str = "all_encoder", normalize_before: bool = False,
                 activation: str = "gelu", position_embedding: str = "learned", **kwargs) -> None:
        super().__init__()
        get = (lambda k, d=None: ablation.get(k, d)) if hasattr(ablation, "get") else (lambda k, d=None: getattr(ablation, k, d))
        unsupported = []
        if arch != "encoder_decoder":
            unsupported.append(f"arch={arch!r} (only 'encoder_decoder', configs/modules/motion_vae.yaml:5)")
        if get("PE_TYPE", "mld") != "mld" or position_embedding != "learned" or get("MLP_DIST", False):
            unsupported.append("only PE_TYPE='mld', learned PE, MLP_DIST=False")
        if normalize_before or activation != "gelu":
            unsupported.append("post-norm / gelu expected")
        if list(latent_dim) != [1, 256] or num_heads * 64 != 256 or num_layers % 2 == 0:
            unsupported.append(f"latent_dim={latent_dim}, num_heads={num_heads}, num_layers={num_layers}")
        if unsupported:
            raise NotImplementedError("HipMldVae: " + "; ".join(unsupported))
        self.latent_size = latent_dim[0]
        self.latent_dim = latent_dim[-1]
        self.nfeats = nfeats
        self.arch = arch
        dims = syn.ModelDims(latent_dim=self.latent_dim, latent_size=self.latent_size, ff_size=ff_size, num_layers=num_layers,
                             num_heads=num_heads, nfeats=nfeats)
        self._register_tree(syn.make_vae_state_dict(seed=1, dims=dims))

    def decode(self, z: torch.Tensor, lengths: List[int]):
        """z [latent_size(=1), B, D], lengths list[int] -> feats [B, max(lengths), nfeats], zeros at padded frames."""
        z = self._check(z, "z")
        lengths = [int(x) for x in lengths]
        if z.dim() != 3 or z.shape[0] != self.latent_size or z.shape[1] != len(lengths) or z.shape[2] != self.latent_dim:
            raise ValueError(f"z must be [{self.latent_size}, {len(lengths)}, {self.latent_dim}], got {tuple(z.shape)}")
        eng = self.sync_weights()
        feats = torch.empty(len(lengths), max(lengths), self.nfeats, dtype=torch.float32, device=z.device)
        eng.vae_decode(z, lengths, feats, self._stream())
        return feats

    def encode(self, features, lengths=None):
        raise NotImplementedError("HipMldVae.encode (mld_vae.py:124-184) is not on the sampling path; it is the next "
                                  "scope row (SURVEY.md §8f.1).  Use the reference MldVae for reconstruction.")

    def forward(self, features, lengths=None):
        raise NotImplementedError("MldVae.forward is a stub in the reference too (mld_vae.py:114-122); use decode().")
