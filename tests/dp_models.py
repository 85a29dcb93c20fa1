"""Shared by tests/dp_worker.py and tests/test_dp_gloo.py: the action (config 5) and diffusion-only (config 4) drop-in models on
the TEST-ONLY simulator engine, small enough for the CPU suite, and the pinned-noise jobs both sides run."""
import os

import numpy as np
import torch

import simlib
from mld_hip import config as C
from mld_hip import engine as E
from mld_hip import synthetic as syn
from mld_hip.datamodule import HipDataModule
from mld_hip.mld import MLD
from mld_hip.text_encoder import SyntheticTextEncoder

STEPS = 2
NOVAE_LAYERS = 2


def state_template(mode):
    if mode == "action":
        sdd, sdv = simlib.action_weights()
        return {**{"denoiser." + k: v for k, v in sdd.items()}, **{"vae." + k: v for k, v in sdv.items()}}
    sd = syn.make_novae_denoiser_state_dict(dims=syn.ModelDims(latent_dim=512, num_layers=NOVAE_LAYERS))
    return {"denoiser." + k: v for k, v in sd.items()}


def build(mode, state, key):
    """(model bound to a fresh simulator engine with `state` loaded through load_state_dict, closer)"""
    if mode == "action":
        eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, **{**simlib.SIM_ACTION_CFG, "max_batch": 4, "max_frames": 16,
                                                                        "num_inference_steps": STEPS})
        E.inject_engine(eng, key)
        cfg = C.load_config(os.path.join(C.CONFIG_DIR, "config_mld_humanact12.yaml"), overrides={"model.scheduler.num_inference_timesteps": STEPS, **simlib.ACTION_OVERRIDES})
        model = MLD(cfg, HipDataModule(cfg, nfeats=150, njoints=25, name="humanact12", engine_key=key), engine_key=key).eval()
    else:
        eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, num_layers=NOVAE_LAYERS,
                                 **{**simlib.NOVAE_CFG, "max_batch": 4, "max_frames": 12, "num_inference_steps": STEPS})
        E.inject_engine(eng, key)
        cfg = C.load_config(os.path.join(C.CONFIG_DIR, "config_novae_humanml3d.yaml"),
                            overrides={"model.scheduler.num_inference_timesteps": STEPS, "model.denoiser.params.num_layers": NOVAE_LAYERS})
        model = MLD(cfg, HipDataModule(cfg, engine_key=key), text_encoder=SyntheticTextEncoder(), engine_key=key).eval()
    missing, unexpected = model.load_state_dict({k: (v if torch.is_tensor(v) else torch.from_numpy(v)) for k, v in state.items()}, strict=False)
    assert not unexpected and all(k.startswith("text_encoder.") for k in missing), (missing[:3], unexpected[:3])

    def close():
        E._engines.pop(key, None)
        eng.close()
    return model, close


def job(mode, n):
    """keyword arguments of DataParallelSampler.__call__ for `n` prompts / labels with per-prompt pinned noise"""
    g = syn._rng(31, "dp_" + mode)
    if mode == "action":
        lengths = [16, 9, 12, 16, 5][:n]
        return dict(actions=[int(a) for a in g.integers(0, 12, n)], lengths=lengths,
                    init_latents=torch.from_numpy(g.standard_normal((n, 1, 256)).astype(np.float32)))
    # equal lengths: in the diffusion-only variant the denoiser attends over the PADDED batch (mld_denoiser.py:208-221 passes no key
    # padding mask to the trans_dec stack), so -- in the reference as here -- a motion depends on the Tmax of the batch it is sampled
    # in; with one Tmax the shard / chunk boundaries cannot matter
    lengths = [12] * n
    return dict(texts=["prompt %d" % i for i in range(n)], lengths=lengths,
                init_latents=torch.from_numpy(g.standard_normal((n, 12, 263)).astype(np.float32)),
                step_noise=torch.from_numpy(g.standard_normal((STEPS, n, 12, 263)).astype(np.float32)))
