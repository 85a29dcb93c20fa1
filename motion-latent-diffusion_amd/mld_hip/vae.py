"""``HipMldVae`` -- drop-in for ``mld.models.architectures.mld_vae.MldVae`` (arch encoder_decoder, PE mld).

``decode(z, lengths)`` (mld_vae.py:186-248) and ``encode(features, lengths)`` (mld_vae.py:124-184; scope row
SURVEY.md §8f.1) run on the HIP engine.  ``encode`` returns ``(latent, Normal(mu, std))`` like the reference;
the N(0,1) draw of ``rsample`` comes from torch's generator on the tensor's device unless ``eps=`` is injected.
"""
from __future__ import annotations

from typing import List, Optional

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
        if list(latent_dim) != [1, 256]:
            unsupported.append("model.latent_dim = %s: only latent_dim: [1, 256] is built.  The reference's latent_dim: [N, 256] ablations (N = 2, 5, 7, 10) make the "
                                   "denoiser attend over N + 2 tokens (mld_denoiser.py:171,187) and the VAE use 2N global tokens / N memory tokens "
                                   "(mld_vae.py:150-163,224-236); the 3-token attention prologue and the 1-key cross-attention shortcut of this engine "
                                   "do not cover them" % (list(latent_dim),))
        elif num_heads * 64 != 256 or num_layers % 2 == 0:
            unsupported.append(f"num_heads={num_heads} (built: 4 heads of 64), num_layers={num_layers} (SkipTransformer needs an odd count)")
        if unsupported:
            raise NotImplementedError("HipMldVae: " + "; ".join(unsupported))
        self.latent_size = latent_dim[0]
        self.latent_dim = latent_dim[-1]
        self.nfeats = nfeats
        self.arch = arch
        dims = syn.ModelDims(latent_dim=self.latent_dim, latent_size=self.latent_size, ff_size=ff_size, num_layers=num_layers,
                             num_heads=num_heads, nfeats=nfeats)
        self._register_tree(syn.make_vae_state_dict(seed=1, dims=dims))
        from . import _lib
        self._set_arch("text", vae_arch=_lib.VAE_MLD, num_layers=int(num_layers), ff_size=int(ff_size), nfeats=int(nfeats))

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

    def encode(self, features: torch.Tensor, lengths: Optional[List[int]] = None, eps: Optional[torch.Tensor] = None):
        """features [B, T, nfeats] (zero padded) -> (latent [latent_size, B, D], torch.distributions.Normal(mu, std))."""
        features = self._check(features, "features")
        if features.dim() != 3 or features.shape[2] != self.nfeats:
            raise ValueError(f"features must be [B, T, {self.nfeats}], got {tuple(features.shape)}")
        B, T = features.shape[0], features.shape[1]
        lengths = [T] * B if lengths is None else [int(x) for x in lengths]      # reference: len(feature) per sample
        if len(lengths) != B or max(lengths) > T:
            raise ValueError("lengths must have one entry per sample and not exceed the padded length")
        eng = self.sync_weights()
        dev = features.device
        if eps is None:
            eps = torch.randn(B, self.latent_dim, device=dev, dtype=torch.float32)   # the draw of Normal.rsample()
        eps = self._check(eps.reshape(B, self.latent_dim), "eps")
        lat = torch.empty(B, self.latent_dim, device=dev)
        mu = torch.empty_like(lat)
        logvar = torch.empty_like(lat)
        eng.vae_encode(features, lengths, T, eps, lat, mu, logvar, self._stream())
        dist = torch.distributions.Normal(mu.unsqueeze(0), logvar.exp().pow(0.5).unsqueeze(0))
        return lat.unsqueeze(0), dist

    def forward(self, features, lengths=None):
        raise NotImplementedError("MldVae.forward is a stub in the reference too (mld_vae.py:114-122); use decode().")


class HipActorVae(HipModule):
    """Drop-in for ``mld.models.architectures.actor_vae.ActorVae`` on the sampling path: ``decode(z, lengths)``
    (actor_vae.py:72-74 -> ActorAgnosticDecoder.forward, :209-235) and ``encode(features, lengths)`` (:64-76 ->
    ActorAgnosticEncoder.forward, :121-175), both on the HIP engine."""
    _prefix = "vae."

    def __init__(self, ablation, nfeats: int, latent_dim: list = [1, 256], ff_size: int = 1024, num_layers: int = 7,
                 num_heads: int = 4, dropout: float = 0.1, is_vae: bool = True, activation: str = "gelu",
                 position_embedding: str = "learned", **kwargs) -> None:
        super().__init__()
        if list(latent_dim) != [1, 256] or num_heads * 64 != 256 or activation != "gelu" or not is_vae:
            raise NotImplementedError(f"HipActorVae: latent_dim={latent_dim}, num_heads={num_heads}, activation={activation!r}, is_vae={is_vae}")
        self.latent_size, self.latent_dim, self.nfeats = latent_dim[0], latent_dim[-1], nfeats
        dims = syn.ModelDims(latent_dim=self.latent_dim, ff_size=ff_size, num_heads=num_heads, nfeats=nfeats)
        self._register_tree(syn.make_actor_vae_state_dict(seed=2, dims=dims, num_layers=num_layers))
        from . import _lib
        self._set_arch("action", vae_arch=_lib.VAE_ACTOR, vae_num_layers=int(num_layers), ff_size=int(ff_size), nfeats=int(nfeats))

    def decode(self, z: torch.Tensor, lengths: List[int]):
        """z [1, B, D], lengths list[int] -> feats [B, max(lengths), nfeats], zeros at padded frames."""
        z = self._check(z, "z")
        lengths = [int(x) for x in lengths]
        if z.dim() != 3 or z.shape[0] != self.latent_size or z.shape[1] != len(lengths) or z.shape[2] != self.latent_dim:
            raise ValueError(f"z must be [{self.latent_size}, {len(lengths)}, {self.latent_dim}], got {tuple(z.shape)}")
        eng = self.sync_weights()
        feats = torch.empty(len(lengths), max(lengths), self.nfeats, dtype=torch.float32, device=z.device)
        eng.vae_decode(z, lengths, feats, self._stream())
        return feats

    def encode(self, features: torch.Tensor, lengths: Optional[List[int]] = None, eps: Optional[torch.Tensor] = None):
        """features [B, T, nfeats] (zero padded) -> (latent [1, B, D], torch.distributions.Normal(mu [B, D], std [B, D]))
        like ActorVae.encode (actor_vae.py:64-76; sample_from_distribution = dist.rsample().unsqueeze(0))."""
        features = self._check(features, "features")
        if features.dim() != 3 or features.shape[2] != self.nfeats:
            raise ValueError(f"features must be [B, T, {self.nfeats}], got {tuple(features.shape)}")
        B, T = features.shape[0], features.shape[1]
        lengths = [T] * B if lengths is None else [int(x) for x in lengths]
        if len(lengths) != B or max(lengths) > T:
            raise ValueError("lengths must have one entry per sample and not exceed the padded length")
        eng = self.sync_weights()
        dev = features.device
        if eps is None:
            eps = torch.randn(B, self.latent_dim, device=dev, dtype=torch.float32)
        eps = self._check(eps.reshape(B, self.latent_dim), "eps")
        lat = torch.empty(B, self.latent_dim, device=dev)
        mu = torch.empty_like(lat)
        logvar = torch.empty_like(lat)
        eng.vae_encode(features, lengths, T, eps, lat, mu, logvar, self._stream())
        return lat.unsqueeze(0), torch.distributions.Normal(mu, logvar.exp().pow(0.5))

    def forward(self, features, lengths=None):
        raise NotImplementedError("ActorVae.forward is a stub in the reference too (actor_vae.py:57-65); use decode().")
