"""Test helper: build + load the TEST-ONLY functional simulator build of libmldhip (tests/hipemu).

The simulator library exports the same C ABI but runs the kernels on host memory through hipsim
(wave64 fibers + an exact model of v_mfma_f32_16x16x4_f32).  It exists so kernel index math can be
checked against the oracle without a GPU.  It is never loaded by the mld_hip package.
"""
import os
import subprocess

import numpy as np

from mld_hip import _lib
from mld_hip import synthetic as syn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "motion-latent-diffusion_amd", "csrc")
SIM_SO = os.path.join(REPO, "tests", "hipemu", "libmldhip_sim.so")

_cached = {}


def sim_library():
    if "lib" not in _cached:
        subprocess.run(["make", "-C", CSRC, "sim"], check=True, capture_output=True)
        _cached["lib"] = _lib.load_library(SIM_SO)
    return _cached["lib"]


def sim_engine(**cfg):
    """Engine on the simulator with the synthetic weights loaded and finalized."""
    eng = _lib.Engine(lib=sim_library(), use_graph=0, **cfg)
    load_synthetic_weights(eng)
    return eng


def load_synthetic_weights(eng, finalize=True):
    ign_d = eng.load_state_dict(syn.make_denoiser_state_dict(), "denoiser.")
    ign_v = eng.load_state_dict(syn.make_vae_state_dict(), "vae.")
    mean, std = syn.make_mean_std()
    eng.load_tensor("mean", mean)
    eng.load_tensor("std", std)
    if finalize:
        eng.finalize()
    return ign_d + ign_v


ACTION_CFG = dict(condition=_lib.COND_ACTION, nclasses=12, vae_arch=_lib.VAE_ACTOR, vae_num_layers=6, num_layers=15, nfeats=150)


def action_weights():
    """(denoiser, ActorVae) synthetic state dicts of the HumanAct12 variant (the fixtures' seeds)."""
    dims = syn.ModelDims(num_layers=15, nfeats=150)
    return (syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12), syn.make_actor_vae_state_dict())


def load_action_weights(eng, finalize=True):
    sdd, sdv = action_weights()
    ign = eng.load_state_dict(sdd, "denoiser.") + eng.load_state_dict(sdv, "vae.")
    if finalize:
        eng.finalize()
    return ign


def sim_action_engine(**cfg):
    eng = _lib.Engine(lib=sim_library(), use_graph=0, **{**ACTION_CFG, **cfg})
    load_action_weights(eng)
    return eng


NOVAE_CFG = dict(latent_dim=512, vae_arch=_lib.VAE_NONE, denoiser_arch=_lib.ARCH_TRANS_DEC, scheduler_type=_lib.SCHED_DDPM,
                 steps_offset=0)


def sim_novae_engine(num_layers=9, **cfg):
    """Diffusion-only engine (config_novae_humanml3d shape, possibly fewer layers / steps) with synthetic weights."""
    eng = _lib.Engine(lib=sim_library(), use_graph=0, num_layers=num_layers, **{**NOVAE_CFG, **cfg})
    eng.load_state_dict(syn.make_novae_denoiser_state_dict(dims=syn.ModelDims(latent_dim=512, num_layers=num_layers)), "denoiser.")
    mean, std = syn.make_mean_std()
    eng.load_tensor("mean", mean)
    eng.load_tensor("std", std)
    eng.finalize()
    return eng
