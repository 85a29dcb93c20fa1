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


# The CPU suite runs the kernels on 3-layer skip stacks (the smallest the engine builds: one input block, the middle block, one
# output block with its skip linear) -- every kernel, fusion and index path of the 9-layer model at a third of the simulator time.
# The full-depth architecture is covered by tests/test_gpu_parity.py (MI355X) and by the fixture tests of the oracle.
SIM_LAYERS = 3
SIM_ACTION_LAYERS, SIM_ACTOR_VAE_LAYERS = 3, 2


def text_weights(num_layers=SIM_LAYERS):
    """(denoiser, VAE) synthetic state dicts of the text model at the simulator's depth"""
    dims = syn.ModelDims(num_layers=num_layers)
    return syn.make_denoiser_state_dict(dims=dims), syn.make_vae_state_dict(dims=dims)


def sim_engine(num_layers=SIM_LAYERS, **cfg):
    """Engine on the simulator with the synthetic weights loaded and finalized."""
    eng = _lib.Engine(lib=sim_library(), use_graph=0, num_layers=num_layers, **cfg)
    load_synthetic_weights(eng, num_layers=num_layers)
    return eng


def load_synthetic_weights(eng, finalize=True, num_layers=None):
    sdd, sdv = text_weights(num_layers if num_layers is not None else eng.cfg.num_layers)
    ign_d = eng.load_state_dict(sdd, "denoiser.")
    ign_v = eng.load_state_dict(sdv, "vae.")
    mean, std = syn.make_mean_std()
    eng.load_tensor("mean", mean)
    eng.load_tensor("std", std)
    if finalize:
        eng.finalize()
    return ign_d + ign_v


ACTION_CFG = dict(condition=_lib.COND_ACTION, nclasses=12, vae_arch=_lib.VAE_ACTOR, vae_num_layers=6, num_layers=15, nfeats=150)   # config_mld_humanact12
SIM_ACTION_CFG = {**ACTION_CFG, "num_layers": SIM_ACTION_LAYERS, "vae_num_layers": SIM_ACTOR_VAE_LAYERS}
ACTION_OVERRIDES = {"model.denoiser.params.num_layers": SIM_ACTION_LAYERS, "model.motion_vae.params.num_layers": SIM_ACTOR_VAE_LAYERS}


def action_weights(num_layers=SIM_ACTION_LAYERS, vae_layers=SIM_ACTOR_VAE_LAYERS):
    """(denoiser, ActorVae) synthetic state dicts of the HumanAct12 variant at the simulator's depth (15 / 6: the fixtures')."""
    dims = syn.ModelDims(num_layers=num_layers, nfeats=150)
    return (syn.make_denoiser_state_dict(seed=3, dims=dims, condition="action", nclasses=12), syn.make_actor_vae_state_dict(num_layers=vae_layers))


def load_action_weights(eng, finalize=True):
    sdd, sdv = action_weights(eng.cfg.num_layers, eng.cfg.vae_num_layers)
    ign = eng.load_state_dict(sdd, "denoiser.") + eng.load_state_dict(sdv, "vae.")
    if finalize:
        eng.finalize()
    return ign


def sim_action_engine(**cfg):
    eng = _lib.Engine(lib=sim_library(), use_graph=0, **{**SIM_ACTION_CFG, **cfg})
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
