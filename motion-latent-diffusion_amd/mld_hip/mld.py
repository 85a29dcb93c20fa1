"""``MLD`` -- the orchestrator of the sampling path with the reference's call surface
(mld/models/modeltype/mld.py:33-143 construct, :216-265 forward, :267-275 gen_from_latent,
:290-360 _diffusion_reverse), minus Lightning/training/metrics (out of scope).

Two execution paths, same results:
  fused   -- every network part is a Hip* drop-in: ONE ``mldhip_sample`` call (hipGraph replay of the
             50-step loop + decode + joints); this is what bench.py measures.
  modular -- the reference's own Python loop over ``denoiser`` / ``scheduler.step`` / ``vae.decode``; used
             when a part was swapped for something else, and by the parity tests of the per-op entry points.
"""
from __future__ import annotations

import inspect
from collections import OrderedDict
from typing import List, Optional

import torch
from torch import nn

from . import engine as _engine
from .config import instantiate_from_config
from .denoiser import HipMldDenoiser
from .scheduler import HipDDIMScheduler, HipDDPMScheduler
from .vae import HipActorVae, HipMldVae


def remove_padding(tensors, lengths):
    """mld/utils/temos_utils.py:24-28."""
    return [t[:n] for t, n in zip(tensors, lengths)]


class MLD(nn.Module):
    def __init__(self, cfg, datamodule, text_encoder: Optional[nn.Module] = None, engine_key: Optional[str] = None, **kwargs):
        super().__init__()
        self.cfg = cfg
        self.stage = cfg.TRAIN.get("STAGE", "diffusion")
        self.condition = cfg.model.condition
        self.nfeats = cfg.DATASET.NFEATS
        self.njoints = cfg.DATASET.NJOINTS
        self.latent_dim = cfg.model.latent_dim
        self.guidance_scale = cfg.model.guidance_scale
        self.datamodule = datamodule
        try:                                                                     # mld.py:50-54
            self.vae_type = cfg.model.vae_type
        except (KeyError, AttributeError):
            self.vae_type = cfg.model.motion_vae.target.split(".")[-1].lower().replace("hip", "").replace("vae", "")
        if self.condition not in ("text", "text_uncond", "action") or self.stage not in ("diffusion", "vae_diffusion"):
            raise NotImplementedError(f"mld_hip.MLD covers text-/action-to-motion sampling (condition={self.condition!r}, stage={self.stage!r})")
        self._engine_key = engine_key
        # engine registry variant ("text_uncond" is the text network fed with empty prompts on both CFG halves, mld.py:228-229)
        self.variant = "novae" if self.vae_type == "no" else ("action" if self.condition == "action" else "text")
        if hasattr(datamodule, "variant") and datamodule.variant is None:
            datamodule.variant = self.variant
        # the reference builds CLIP for every condition (mld.py:60); the action path never calls it, so it is skipped there
        self.text_encoder = text_encoder if (text_encoder is not None or self.condition == "action") \
            else instantiate_from_config(cfg.model.text_encoder)
        self.vae = instantiate_from_config(cfg.model.motion_vae) if self.vae_type != "no" else None      # mld.py:58-59
        self.denoiser = instantiate_from_config(cfg.model.denoiser)
        self.scheduler = instantiate_from_config(cfg.model.scheduler)
        for m in (self.vae, self.denoiser):
            if engine_key is not None and m is not None and hasattr(m, "use_engine"):
                m.use_engine(engine_key)
        # ONE engine for all parts of this model: every part asks the registry for the union of the architecture fields
        # (the fused sample() needs every weight group in one handle); another model with other fields gets its own engine
        shared = {}
        for m in (self.denoiser, self.vae):
            shared.update(getattr(m, "_arch", {}) or {})
        if hasattr(self.scheduler, "engine_config"):
            shared.update(self.scheduler.engine_config(cfg.model.scheduler.num_inference_timesteps))
        shared["guidance_scale"] = float(self.guidance_scale)
        for m in (self.denoiser, self.vae, datamodule, self.scheduler):
            if m is not None and hasattr(m, "_shared_arch") and engine_key is None:
                m._shared_arch = shared
        if hasattr(self.scheduler, "_variant"):
            self.scheduler._variant = self.variant
        self.sample_mean = False
        self.fact = None
        self.do_classifier_free_guidance = self.guidance_scale > 1.0
        self.feats2joints = datamodule.feats2joints
        self.times: List[float] = []

    # ------------------------------------------------------------------ checkpoint contract (base.py:96-127)
    def load_state_dict(self, state_dict, strict: bool = True):
        te = self.text_encoder.state_dict() if self.text_encoder is not None else {}
        new = OrderedDict(("text_encoder." + k, v) for k, v in te.items())
        for k, v in state_dict.items():
            if "text_encoder" not in k and not k.startswith("t2m_"):      # evaluator nets are not part of sampling
                new[k] = v
        return super().load_state_dict(new, strict)

    # ------------------------------------------------------------------ helpers
    @property
    def fused(self) -> bool:
        if self.vae_type == "no":
            return (isinstance(self.denoiser, HipMldDenoiser) and isinstance(self.scheduler, HipDDPMScheduler)
                    and self.do_classifier_free_guidance)
        return (isinstance(self.denoiser, HipMldDenoiser) and isinstance(self.vae, (HipMldVae, HipActorVae))
                and isinstance(self.scheduler, HipDDIMScheduler) and self.do_classifier_free_guidance)

    def _engine(self):
        eng = self.denoiser.sync_weights()
        if self.vae is not None:
            self.vae.sync_weights()
        want = self.scheduler.engine_config(self.cfg.model.scheduler.num_inference_timesteps)
        for k in ("num_train_timesteps", "num_inference_steps", "steps_offset", "set_alpha_to_one"):
            if getattr(eng.cfg, k) != want[k]:
                raise RuntimeError(f"engine/scheduler mismatch on {k}: engine {getattr(eng.cfg, k)}, scheduler {want[k]}; "
                                   "create the engine with mld_hip.engine.configure(**scheduler.engine_config(n)) first")
        if abs(eng.cfg.guidance_scale - self.guidance_scale) > 1e-6:
            raise RuntimeError("engine guidance_scale differs from cfg.model.guidance_scale")
        return eng

    # ------------------------------------------------------------------ fused path
    @torch.no_grad()
    def sample(self, text_emb: torch.Tensor, lengths: List[int], init_latents: Optional[torch.Tensor] = None):
        """text_emb [2B, 1, 768] (uncond half first) -> (joints [B,T,22,3], feats [B,T,nfeats], latents [B,1,D]) on device."""
        lengths = [int(x) for x in lengths]
        B, T = len(lengths), max(lengths)
        dev = text_emb.device
        text_emb = text_emb.float().contiguous()
        if init_latents is None:
            init_latents = torch.randn((B, self.latent_dim[0], self.latent_dim[-1]), device=dev, dtype=torch.float)   # mld.py:303
        init_latents = init_latents.float().contiguous()
        eng = self._engine()
        dm_eng = self.datamodule._engine(dev) if hasattr(self.datamodule, "_engine") else eng
        if dm_eng is not eng:
            raise RuntimeError("datamodule and network parts are bound to different engines")
        _engine.finalize_if_dirty(eng, _engine.current_stream_handle(text_emb))
        lat = torch.empty(B, self.latent_dim[0], self.latent_dim[-1], device=dev)
        feats = torch.empty(B, T, self.nfeats, device=dev)
        joints = torch.empty(B, T, self.njoints, 3, device=dev)
        eng.sample(text_emb, init_latents, lengths, lat, feats, joints, _engine.current_stream_handle(text_emb))
        return joints, feats, lat

    @torch.no_grad()
    def sample_many(self, requests, init_latents=None, pipeline: bool = False):
        """Several independent text-to-motion requests as ONE engine call (``mldhip_sample_many``: one reverse-diffusion chain +
        one decode over all of them; the engine needs ``max_batch >= total motions``).  `requests` = [(text_emb [2B_i,1,768],
        lengths_i), ...]; returns [(joints_i, feats_i, latents_i), ...] on device, each shaped as ``sample`` would return it.

        ``pipeline=True`` (engine option "many_pipeline"; ``mld_hip.engine.configure("text", max_in_flight=2)`` before the first use): the
        requests run ONE AFTER THE OTHER -- the reference's own loop, batch after batch (mld.py:618-672) -- each exactly what ``sample`` returns
        for it, with the decode of request k overlapped with the reverse loop of request k + 1 (bs-64 requests: 7.0 instead of 8.0 ms each)."""
        if self.vae_type == "no" or self.condition == "action":
            raise NotImplementedError("sample_many serves the text-to-motion latent model")
        eng = self._engine()
        dev = requests[0][0].device
        dm_eng = self.datamodule._engine(dev) if hasattr(self.datamodule, "_engine") else eng
        if dm_eng is not eng:
            raise RuntimeError("datamodule and network parts are bound to different engines")
        stream = _engine.current_stream_handle(requests[0][0])
        _engine.finalize_if_dirty(eng, stream)
        reqs, outs, keep = [], [], []
        for i, (text_emb, lengths) in enumerate(requests):
            lengths = [int(x) for x in lengths]
            B, T = len(lengths), max(lengths)
            text_emb = text_emb.float().contiguous()
            lat0 = init_latents[i] if init_latents is not None else torch.randn((B, self.latent_dim[0], self.latent_dim[-1]), device=dev)
            lat0 = lat0.float().contiguous()
            lat = torch.empty(B, self.latent_dim[0], self.latent_dim[-1], device=dev)
            feats = torch.empty(B, T, self.nfeats, device=dev)
            joints = torch.empty(B, T, self.njoints, 3, device=dev)
            keep.append((text_emb, lat0))
            reqs.append(dict(text_emb=text_emb, init_latents=lat0, lengths=lengths, latents_out=lat, feats_out=feats, joints_out=joints))
            outs.append((joints, feats, lat))
        if pipeline:
            eng.set_option("many_pipeline", 1)
        try:
            eng.sample_many(reqs, stream)
        finally:
            if pipeline:
                eng.set_option("many_pipeline", 0)
        return outs

    @torch.no_grad()
    def sample_novae(self, text_emb: torch.Tensor, lengths: List[int], init_latents: Optional[torch.Tensor] = None,
                     step_noise: Optional[torch.Tensor] = None, seed: Optional[int] = None):
        """Diffusion-only sampling in ONE mldhip_sample_novae call: text_emb [2B, 1, 768] -> (joints [B,T,22,3], feats [B,T,nfeats]).
        step_noise [steps, B, T, nfeats] injects the scheduler's per-step draws (parity runs); otherwise they come from the
        engine's Philox stream keyed by `seed` (default: drawn from torch's generator, so torch.manual_seed governs it)."""
        lengths = [int(x) for x in lengths]
        B, T = len(lengths), max(lengths)
        dev = text_emb.device
        text_emb = text_emb.float().contiguous()
        if init_latents is None:
            init_latents = torch.randn((B, T, self.nfeats), device=dev, dtype=torch.float)            # mld.py:296-301
        init_latents = init_latents.float().contiguous()
        if step_noise is not None:
            step_noise = step_noise.float().contiguous()
            if tuple(step_noise.shape) != (self.cfg.model.scheduler.num_inference_timesteps, B, T, self.nfeats):
                raise ValueError(f"step_noise must be [steps, B, T, nfeats], got {tuple(step_noise.shape)}")
        elif seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        eng = self._engine()
        if hasattr(self.datamodule, "_engine") and self.datamodule._engine(dev) is not eng:
            raise RuntimeError("datamodule and denoiser are bound to different engines")
        _engine.finalize_if_dirty(eng, _engine.current_stream_handle(text_emb))
        feats = torch.empty(B, T, self.nfeats, device=dev)
        joints = torch.empty(B, T, self.njoints, 3, device=dev)
        eng.sample_novae(text_emb, init_latents, lengths, step_noise, seed or 0, feats, joints, _engine.current_stream_handle(text_emb))
        return joints, feats

    @torch.no_grad()
    def sample_action(self, actions, lengths: List[int], init_latents: Optional[torch.Tensor] = None, device=None):
        """Action labels [B] / [B, 1] -> (feats [B, T, nfeats], latents [B, 1, D]) on device: ONE mldhip_sample_action call."""
        lengths = [int(x) for x in lengths]
        acts = [int(a) for a in (actions.reshape(-1).tolist() if torch.is_tensor(actions) else list(actions))]
        B, T = len(lengths), max(lengths)
        dev = init_latents.device if init_latents is not None else (device or next(self.denoiser.parameters()).device)
        if init_latents is None:
            init_latents = torch.randn((B, self.latent_dim[0], self.latent_dim[-1]), device=dev, dtype=torch.float)   # mld.py:303
        init_latents = init_latents.float().contiguous()
        eng = self._engine()
        lat = torch.empty(B, self.latent_dim[0], self.latent_dim[-1], device=dev)
        feats = torch.empty(B, T, self.nfeats, device=dev)
        eng.sample_action(acts, init_latents, lengths, lat, feats, _engine.current_stream_handle(init_latents))
        return feats, lat

    @torch.no_grad()
    def sample_many_action(self, requests, init_latents=None, device=None):
        """Several action-to-motion requests as ONE engine call (``mldhip_sample_many`` with ``actions_host``): `requests` =
        [(actions_i, lengths_i), ...] -> [(feats_i [B_i, T_i, nfeats], latents_i), ...] on device; the engine needs
        ``max_batch >= total motions`` (four bs-256 requests per call: 30.0 k motions/s against 5.9 k one call at a time, bench.py)."""
        if self.condition != "action":
            raise NotImplementedError("sample_many_action serves the action-conditioned model")
        eng = self._engine()
        dev = device or (init_latents[0].device if init_latents is not None else next(self.denoiser.parameters()).device)
        _engine.finalize_if_dirty(eng)
        reqs, outs, keep = [], [], []
        for i, (actions, lengths) in enumerate(requests):
            lengths = [int(x) for x in lengths]
            acts = [int(a) for a in (actions.reshape(-1).tolist() if torch.is_tensor(actions) else list(actions))]
            B, T = len(lengths), max(lengths)
            lat0 = init_latents[i] if init_latents is not None else torch.randn((B, self.latent_dim[0], self.latent_dim[-1]), device=dev)
            lat0 = lat0.to(dev).float().contiguous()
            lat = torch.empty(B, self.latent_dim[0], self.latent_dim[-1], device=dev)
            feats = torch.empty(B, T, self.nfeats, device=dev)
            keep.append(lat0)
            reqs.append(dict(actions=acts, init_latents=lat0, lengths=lengths, latents_out=lat, feats_out=feats))
            outs.append((feats, lat))
        eng.sample_many(reqs, _engine.current_stream_handle(keep[0]))
        return outs

    @torch.no_grad()
    def a2m_eval(self, batch, init_latents: Optional[torch.Tensor] = None):
        """Sampling core of MLD.a2m_eval (mld.py:710-735): batch["action"] [B, 1] labels, batch["length"] -> rs_set with
        ``m_action`` / ``m_rst`` (features [B, T, nfeats]) / ``m_lens``.  The joints entries of the reference's rs_set go
        through SMPL (mld/transforms/rots2joints/smplh.py) and are not produced here."""
        actions, lengths = batch["action"], list(batch["length"])
        if self.fused:
            feats, _ = self.sample_action(actions, lengths, init_latents,
                                          device=actions.device if torch.is_tensor(actions) and actions.is_cuda else None)
        else:
            a = actions if torch.is_tensor(actions) else torch.tensor(actions)
            cond = torch.cat((torch.zeros_like(a), a)) if self.do_classifier_free_guidance else a      # mld.py:716-717
            z = self._diffusion_reverse(cond.reshape(-1, 1), lengths, init_latents)
            feats = self.vae.decode(z.contiguous(), lengths)
        return {"m_action": actions, "m_rst": feats, "m_lens": lengths}

    # ------------------------------------------------------------------ reference surface
    @torch.no_grad()
    def forward(self, batch, init_latents: Optional[torch.Tensor] = None, step_noise: Optional[torch.Tensor] = None):
        texts, lengths = list(batch["text"]), list(batch["length"])
        if self.do_classifier_free_guidance:                                    # mld.py:224-230: uncond half first
            texts = [""] * len(texts) + ([""] * len(texts) if self.condition == "text_uncond" else texts)
        text_emb = self.text_encoder(texts)
        if self.vae_type == "no":
            if self.fused:
                joints, _ = self.sample_novae(text_emb, lengths, init_latents, step_noise)
            else:
                z = self._diffusion_reverse(text_emb, lengths, init_latents, step_noise)
                joints = self.feats2joints(z.permute(1, 0, 2).contiguous())     # mld.py:241-242: "decode" is a permute
            return remove_padding(joints.detach().cpu(), lengths)
        if self.fused:
            joints, _, _ = self.sample(text_emb, lengths, init_latents)
        else:
            z = self._diffusion_reverse(text_emb, lengths, init_latents)
            feats = self.vae.decode(z, lengths)
            joints = self.feats2joints(feats.detach())
        return remove_padding(joints.detach().cpu(), lengths)                   # mld.py:264-265

    @torch.no_grad()
    def gen_from_latent(self, batch):
        feats = self.vae.decode(batch["latent"], batch["length"])
        return remove_padding(self.feats2joints(feats.detach()).cpu(), batch["length"])

    @torch.no_grad()
    def recon_from_motion(self, batch):
        """mld.py:277-288: encode -> decode -> joints for the reconstruction and the reference motion."""
        feats_ref, length = batch["motion"], list(batch["length"])
        z, _ = self.vae.encode(feats_ref, length)
        feats_rst = self.vae.decode(z.contiguous(), length)
        joints = self.feats2joints(feats_rst.detach())
        joints_ref = self.feats2joints(feats_ref.detach().contiguous())
        return remove_padding(joints.cpu(), length), remove_padding(joints_ref.cpu(), length)

    @torch.no_grad()
    def _diffusion_reverse(self, encoder_hidden_states, lengths=None, init_latents: Optional[torch.Tensor] = None,
                           step_noise: Optional[torch.Tensor] = None):
        """The reference's Python loop (mld.py:290-360) over the drop-in parts -> [latent_size, B, D]
        ([T, B, nfeats] for vae_type 'no')."""
        bsz = encoder_hidden_states.shape[0] // (2 if self.do_classifier_free_guidance else 1)
        dev = init_latents.device if init_latents is not None else encoder_hidden_states.device
        if self.vae_type == "no":
            assert lengths is not None, "no vae (diffusion only) need lengths for diffusion"          # mld.py:295
            shape = (bsz, max(lengths), self.nfeats)
        else:
            shape = (bsz, self.latent_dim[0], self.latent_dim[-1])
        latents = init_latents if init_latents is not None else torch.randn(shape, device=dev, dtype=torch.float)
        latents = latents * self.scheduler.init_noise_sigma
        self.scheduler.set_timesteps(self.cfg.model.scheduler.num_inference_timesteps)
        extra = {}
        if "eta" in set(inspect.signature(self.scheduler.step).parameters.keys()):
            extra["eta"] = self.cfg.model.scheduler.eta
        takes_noise = "noise" in set(inspect.signature(self.scheduler.step).parameters.keys())
        for i, t in enumerate(self.scheduler.timesteps.tolist()):
            if step_noise is not None and takes_noise:
                extra["noise"] = step_noise[i]
            x = torch.cat([latents] * 2) if self.do_classifier_free_guidance else latents
            noise_pred = self.denoiser(sample=x, timestep=t, encoder_hidden_states=encoder_hidden_states,
                                       lengths=(list(lengths) * 2 if lengths is not None else None))[0]
            if self.do_classifier_free_guidance:
                u, c = noise_pred.chunk(2)
                noise_pred = u + self.guidance_scale * (c - u)
            latents = self.scheduler.step(noise_pred, t, latents, **extra).prev_sample
        return latents.permute(1, 0, 2)
