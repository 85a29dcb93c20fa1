"""``HipDDIMScheduler`` -- drop-in for ``diffusers.DDIMScheduler`` as the reference uses it
(configs/modules/scheduler.yaml:1-14; call sites mld.py:81-83,310-320,345-346).

Third-party arithmetic restated from the published algorithm (diffusers is not installed; SURVEY.md App.
A.3, parity unpinned): float32 tables, scaled_linear betas, steps_offset, set_alpha_to_one=False, eta=0.
The tables are host logic.  ``step`` on device tensors goes through the C ABI (``mldhip_ddim_step`` / ``mldhip_ddpm_step`` of the
engine of that device, whose tables are checked against this scheduler's fields by the registry); on CPU tensors it is the same
four elementwise operations in torch (host logic, used by the CPU tests).  The fused ``MLD.sample`` never calls ``step``: the same
coefficients are computed inside libmldhip and applied by the step-final kernel.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


class SchedulerOutput(SimpleNamespace):
    pass


class HipDDIMScheduler:
    _variant = "text"        # engine registry variant / architecture fields of the model this scheduler serves (set by MLD)
    _shared_arch: dict = {}

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", clip_sample: bool = True, set_alpha_to_one: bool = True,
                 steps_offset: int = 0, prediction_type: str = "epsilon", **kwargs):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(f"beta_schedule={beta_schedule!r}: the MLD configs use 'scaled_linear'")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not used by any MLD config")
        if prediction_type != "epsilon":
            raise NotImplementedError("PREDICT_EPSILON=False (prediction_type='sample') is not shipped by any MLD config")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                                      steps_offset=steps_offset, prediction_type=prediction_type)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def engine_config(self, num_inference_steps: int):
        """Fields of mldhip_config that must agree with this scheduler for the fused path."""
        c = self.config
        return dict(num_train_timesteps=c.num_train_timesteps, num_inference_steps=num_inference_steps,
                    steps_offset=c.steps_offset, set_alpha_to_one=int(c.set_alpha_to_one),
                    beta_start=c.beta_start, beta_end=c.beta_end)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta: float = 0.0, **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta > 0 draws noise inside the scheduler; MLD uses eta = 0 (scheduler.yaml:4)")
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps() first")
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        sa, sb = float(a_t ** 0.5), float((1 - a_t) ** 0.5)
        pa, pb = float(a_p ** 0.5), float((1 - a_p) ** 0.5)
        x0 = (sample - sb * model_output) / sa
        # device tensors: the step runs in libmldhip (mldhip_ddim_step) -- but only on the engine of the model this scheduler
        # belongs to, and only while set_timesteps() agrees with that engine's grid: mldhip_ddim_step derives t_prev from the
        # ENGINE's num_inference_steps.  A stand-alone scheduler, or one re-gridded with another step count, uses the torch
        # arithmetic below (same formula) instead of instantiating an engine for an elementwise update.
        if (sample.is_cuda and sample.dtype == torch.float32 and model_output.dtype == torch.float32 and self._shared_arch
                and self._shared_arch.get("num_inference_steps") == self.num_inference_steps):
            from . import engine as _engine
            eng = _engine.get_engine(sample.device, self._variant, want=self._shared_arch)
            out = torch.empty_like(sample, memory_format=torch.contiguous_format)
            eng.ddim_step(model_output.contiguous(), t, sample.contiguous(), out, out.numel(), _engine.current_stream_handle(sample))
            return SchedulerOutput(prev_sample=out, pred_original_sample=x0)
        return SchedulerOutput(prev_sample=pa * x0 + pb * model_output, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod.to(original_samples.device)[timesteps].reshape(-1, *([1] * (original_samples.dim() - 1)))
        return a.sqrt() * original_samples + (1 - a).sqrt() * noise


class HipDDPMScheduler:
    """Drop-in for ``diffusers.DDPMScheduler`` as configs/modules_novae/scheduler.yaml:16-29 uses it (variance_type
    fixed_small, no clipping, epsilon prediction).  Third-party arithmetic restated from the published algorithm
    (PARITY UNPINNED).  ``step`` has no ``eta`` parameter -- the reference probes for it
    (mld.py:318-320) -- and draws its noise from torch's generator unless ``noise=`` is injected."""

    _variant = "novae"
    _shared_arch: dict = {}

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", variance_type: str = "fixed_small", clip_sample: bool = True,
                 prediction_type: str = "epsilon", **kwargs):
        if beta_schedule != "scaled_linear" or variance_type != "fixed_small" or clip_sample or prediction_type != "epsilon":
            raise NotImplementedError("HipDDPMScheduler: scaled_linear / fixed_small / clip_sample=False / epsilon only "
                                      "(configs/modules_novae/scheduler.yaml)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                                      prediction_type=prediction_type)
        f = np.float32      # numpy float32, sequential cumprod: bit-identical to the engine's host tables
        betas = (np.linspace(f(beta_start) ** f(0.5), f(beta_end) ** f(0.5), num_train_timesteps, dtype=f) ** 2).astype(f)
        self.betas = torch.from_numpy(betas)
        self.alphas = torch.from_numpy((f(1.0) - betas).astype(f))
        self.alphas_cumprod = torch.from_numpy(np.cumprod(self.alphas.numpy(), dtype=f))
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def engine_config(self, num_inference_steps: int):
        c = self.config
        return dict(num_train_timesteps=c.num_train_timesteps, num_inference_steps=num_inference_steps, steps_offset=0,
                    set_alpha_to_one=0, beta_start=c.beta_start, beta_end=c.beta_end)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def coeffs(self, t: int):
        f = np.float32
        ratio = self.config.num_train_timesteps // (self.num_inference_steps or self.config.num_train_timesteps)
        acp = self.alphas_cumprod.numpy()
        ab_t, ab_p = acp[t], (acp[t - ratio] if t - ratio >= 0 else f(1.0))
        if ratio == 1:      # table values, as the diffusers releases contemporary with the reference read them
            a_t, b_t = self.alphas.numpy()[t], self.betas.numpy()[t]
        else:
            a_t = f(ab_t / ab_p)
            b_t = f(f(1.0) - a_t)
        bp_t, bp_p = f(f(1.0) - ab_t), f(f(1.0) - ab_p)
        var = f(max(float(bp_p / bp_t * b_t), 1e-20))
        return (float(np.sqrt(ab_t, dtype=f)), float(np.sqrt(bp_t, dtype=f)), float(f(np.sqrt(ab_p, dtype=f) * b_t / bp_t)),
                float(f(np.sqrt(a_t, dtype=f) * bp_p / bp_t)), float(np.sqrt(var, dtype=f)) if t > 0 else 0.0)

    def step(self, model_output, timestep, sample, generator=None, noise=None, **kwargs):
        sa, sb, c0, c1, sg = self.coeffs(int(timestep))
        x0 = (sample - sb * model_output) / sa
        if sg != 0.0 and noise is None:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
        if sample.is_cuda and sample.dtype == torch.float32 and model_output.dtype == torch.float32 \
                and (self.num_inference_steps or self.config.num_train_timesteps) == self.config.num_train_timesteps:
            from . import _lib, engine as _engine     # device tensors at the shipped setting (1000 steps): mldhip_ddpm_step
            want = dict(self.engine_config(self.config.num_train_timesteps), vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC,
                        scheduler_type=_lib.SCHED_DDPM, latent_dim=512)
            eng = _engine.get_engine(sample.device, "novae", want=self._shared_arch or want)
            out = torch.empty_like(sample, memory_format=torch.contiguous_format)
            nz = noise.contiguous() if noise is not None else torch.zeros_like(out)      # t = 0: sigma is 0, the draw is unused
            eng.ddpm_step(model_output.contiguous(), int(timestep), sample.contiguous(), nz, out, out.numel(),
                          stream=_engine.current_stream_handle(sample))
            return SchedulerOutput(prev_sample=out, pred_original_sample=x0)
        prev = c0 * x0 + c1 * sample
        if sg != 0.0:
            prev = prev + sg * noise
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod.to(original_samples.device)[timesteps].reshape(-1, *([1] * (original_samples.dim() - 1)))
        return a.sqrt() * original_samples + (1 - a).sqrt() * noise
