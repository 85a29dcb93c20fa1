"""``HipMldDenoiser`` -- drop-in for ``mld.models.architectures.mld_denoiser.MldDenoiser`` (text and action conditions on
VAE latents with the skip trans_enc; text condition on raw motion with trans_dec, i.e. VAE_TYPE 'no').

Same constructor keywords (mld_denoiser.py:18-38), same ``forward(sample, timestep,
encoder_hidden_states, lengths=None) -> (sample,)`` contract (mld_denoiser.py:135-228), same
``state_dict`` keys; the arithmetic runs in libmldhip's HIP kernels.
"""
from __future__ import annotations

import torch

from . import synthetic as syn
from ._module import HipModule


class HipMldDenoiser(HipModule):
    _prefix = "denoiser."

    def __init__(self, ablation, nfeats: int = 263, condition: str = "text", latent_dim: list = [1, 256],
                 ff_size: int = 1024, num_layers: int = 6, num_heads: int = 4, dropout: float = 0.1,
                 normalize_before: bool = False, activation: str = "gelu", flip_sin_to_cos: bool = True,
                 return_intermediate_dec: bool = False, position_embedding: str = "learned", arch: str = "trans_enc",
                 freq_shift: int = 0, guidance_scale: float = 7.5, guidance_uncondp: float = 0.1,
                 text_encoded_dim: int = 768, nclasses: int = 10, **kwargs) -> None:
        super().__init__()
        abl = ablation if isinstance(ablation, dict) else vars(ablation) if not hasattr(ablation, "get") else ablation
        get = (lambda k, d=None: abl.get(k, d)) if hasattr(abl, "get") else (lambda k, d=None: getattr(ablation, k, d))
        unsupported = []
        if condition not in ("text", "text_uncond", "action"):
            unsupported.append(f"condition={condition!r} (the reference knows text, text_uncond, action; mld_denoiser.py:57-79)")
        novae = get("VAE_TYPE", "mld") == "no"
        if novae != (arch == "trans_dec") or arch not in ("trans_enc", "trans_dec"):
            unsupported.append(f"arch={arch!r} with VAE_TYPE={get('VAE_TYPE')!r} (built: trans_enc on VAE latents, trans_dec on raw motion)")
        if not novae and not get("SKIP_CONNECT", False):
            unsupported.append("SKIP_CONNECT=False (only the skip trans_enc of the shipped configs)")
        if novae and condition not in ("text", "text_uncond"):
            unsupported.append("the diffusion-only variant is text-conditioned (config_novae_humanml3d.yaml)")
        if get("DIFF_PE_TYPE", "mld") != "mld" or position_embedding != "learned":
            unsupported.append("only DIFF_PE_TYPE='mld' with learned positional embeddings")
        if normalize_before or activation != "gelu" or not flip_sin_to_cos or freq_shift != 0:
            unsupported.append("post-norm / gelu / flip_sin_to_cos=True / freq_shift=0 expected")
        if novae:
            if list(latent_dim) != [1, 512] or num_heads * 128 != 512:
                unsupported.append(f"diffusion-only variant: latent_dim={latent_dim}, num_heads={num_heads} (built: [1, 512], 4 heads)")
        elif list(latent_dim) != [1, 256]:
            unsupported.append("model.latent_dim = %s: only latent_dim: [1, 256] is built.  The reference's latent_dim: [N, 256] ablations (N = 2, 5, 7, 10) make the "
                                   "denoiser attend over N + 2 tokens (mld_denoiser.py:171,187) and the VAE use 2N global tokens / N memory tokens "
                                   "(mld_vae.py:150-163,224-236); the 3-token attention prologue and the 1-key cross-attention shortcut of this engine "
                                   "do not cover them" % (list(latent_dim),))
        elif num_heads * 64 != 256 or num_layers % 2 == 0:
            unsupported.append(f"num_heads={num_heads} (built: 4 heads of 64), num_layers={num_layers} (SkipTransformer needs an odd count)")
        if unsupported:
            raise NotImplementedError("HipMldDenoiser: " + "; ".join(unsupported))
        self.diffusion_only = novae
        self.nfeats = nfeats
        self.latent_dim = latent_dim[-1]
        self.text_encoded_dim = text_encoded_dim
        self.condition = condition
        self.arch = arch
        self.num_layers = num_layers
        self.ff_size = ff_size
        self.nclasses = nclasses
        self.guidance_scale = guidance_scale
        dims = syn.ModelDims(latent_dim=self.latent_dim, latent_size=latent_dim[0], ff_size=ff_size, num_layers=num_layers,
                             num_heads=num_heads, nfeats=nfeats, text_dim=text_encoded_dim)
        from . import _lib
        if novae:
            self._register_tree(syn.make_novae_denoiser_state_dict(seed=4, dims=dims))
            self._set_arch("novae", vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC, latent_dim=self.latent_dim,
                           num_layers=int(num_layers), ff_size=int(ff_size), text_dim=int(text_encoded_dim), nfeats=int(nfeats))
            return
        self._register_tree(syn.make_denoiser_state_dict(seed=0, dims=dims, condition=condition, nclasses=nclasses))
        if condition == "action":
            self._set_arch("action", condition=_lib.COND_ACTION, nclasses=int(nclasses), num_layers=int(num_layers),
                           ff_size=int(ff_size), guidance_scale=float(guidance_scale))
        else:
            self._set_arch("text", condition=_lib.COND_TEXT, num_layers=int(num_layers), ff_size=int(ff_size),
                           text_dim=int(text_encoded_dim))

    def forward(self, sample, timestep, encoder_hidden_states, lengths=None, **kwargs):
        """sample [R, 1, D], timestep int / 0-d tensor, encoder_hidden_states [R, 1, text_dim] (text) or [R, 1] class
        labels (action; the first R/2 rows are the unconditional half when guidance_scale > 1) -> ([R, 1, D],)."""
        sample = self._check(sample, "sample")
        if self.diffusion_only:
            # raw-motion rows [R, T, nfeats]; lengths are mandatory here (mld_denoiser.py:144-146,219-221)
            text = self._check(encoder_hidden_states, "encoder_hidden_states")
            if sample.dim() != 3 or sample.shape[2] != self.nfeats:
                raise ValueError(f"sample must be [R, T, {self.nfeats}], got {tuple(sample.shape)}")
            if lengths in (None, []) or len(lengths) != sample.shape[0]:
                raise ValueError("the diffusion-only denoiser needs one length per row of sample")
            if text.shape[0] != sample.shape[0] or text.numel() != sample.shape[0] * self.text_encoded_dim:
                raise ValueError(f"encoder_hidden_states must be [R, 1, {self.text_encoded_dim}], got {tuple(text.shape)}")
            t = int(timestep.reshape(-1)[0].item()) if torch.is_tensor(timestep) else int(timestep)
            eng = self.sync_weights()
            out = torch.empty_like(sample)
            eng.denoiser_forward_novae(sample, t, text, [int(x) for x in lengths], sample.shape[1], out, self._stream())
            return (out,)
        if sample.dim() != 3 or sample.shape[1] != 1 or sample.shape[2] != self.latent_dim:
            raise ValueError(f"sample must be [R, 1, {self.latent_dim}], got {tuple(sample.shape)}")
        if self.condition == "action":
            # EmbedAction: idx = input[:, 0].long() (mld_denoiser.py:250); labels cross to the host like lengths do
            acts = encoder_hidden_states.reshape(encoder_hidden_states.shape[0], -1)[:, 0].long().cpu().tolist()
            if len(acts) != sample.shape[0]:
                raise ValueError(f"encoder_hidden_states must hold one action label per row of sample ({sample.shape[0]}), got {len(acts)}")
            t = int(timestep.reshape(-1)[0].item()) if torch.is_tensor(timestep) else int(timestep)
            eng = self.sync_weights()
            out = torch.empty_like(sample)
            eng.denoiser_forward_action(sample, t, acts, out, self._stream())
            return (out,)
        text = self._check(encoder_hidden_states, "encoder_hidden_states")
        if text.shape[0] != sample.shape[0] or text.shape[-1] != self.text_encoded_dim or text.numel() != sample.shape[0] * self.text_encoded_dim:
            raise ValueError(f"encoder_hidden_states must be [R, 1, {self.text_encoded_dim}], got {tuple(text.shape)}")
        t = int(timestep.reshape(-1)[0].item()) if torch.is_tensor(timestep) else int(timestep)
        eng = self.sync_weights()
        out = torch.empty_like(sample)
        eng.denoiser_forward(sample, t, text, sample.shape[0], out, self._stream())
        return (out,)
