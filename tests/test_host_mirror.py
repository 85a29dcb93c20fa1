"""Host-side mirror of the reference's plugin surface (config merge, drop-in modules, MLD orchestrator),
exercised on CPU through the TEST-ONLY simulator engine (injected; the package itself has no CPU path)."""
import json
import os

import numpy as np
import pytest
import torch

import simlib
from mld_hip import config as C
from mld_hip import engine as E
from mld_hip import synthetic as syn
from mld_hip.datamodule import HipDataModule
from mld_hip.denoiser import HipMldDenoiser
from mld_hip.mld import MLD
from mld_hip.scheduler import HipDDIMScheduler, HipDDPMScheduler
from mld_hip.text_encoder import SyntheticTextEncoder
from mld_hip.vae import HipActorVae, HipMldVae
from oracle import mld_oracle as O

# the simulator engines of this file run 3-layer skip stacks (tests/simlib.py): the YAML's layer counts are overridden to match
TEXT_OVERRIDES = {"model.denoiser.params.num_layers": simlib.SIM_LAYERS, "model.motion_vae.params.num_layers": simlib.SIM_LAYERS}

REF = "/root/reference"


# ------------------------------------------------------------------------------------------ config
def test_config_merge_order_and_interpolation():
    cfg = C.load_config()
    assert cfg.model.target == "modules_hip"
    d = cfg.model.denoiser
    assert d.target == "mld_hip.denoiser.HipMldDenoiser"
    assert d.params.latent_dim == [1, 256] and d.params.condition == "text" and d.params.guidance_scale == 7.5
    assert d.params.nfeats == 263 and d.params.num_layers == 9
    abl = d.params.ablation
    assert abl.SKIP_CONNECT is True and abl.PE_TYPE == "mld" and abl.DIFF_PE_TYPE == "mld"   # experiment overrides base
    assert abl.PREDICT_EPSILON is True and abl.MLP_DIST is False                             # base survives
    assert cfg.model.scheduler.num_inference_timesteps == 50 and cfg.model.scheduler.eta == 0.0
    assert cfg.model.scheduler.params.steps_offset == 1 and cfg.model.scheduler.params.set_alpha_to_one is False
    assert cfg.model.text_encoder.params.modelpath == cfg.model.clip_path                     # from assets.yaml
    over = C.load_config(overrides={"model.guidance_scale": 3.0})
    assert over.model.denoiser.params.guidance_scale == 3.0                                   # interpolation sees override


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_hot_path_params_equal_the_reference_yamls():
    """Same `params:` blocks as configs/modules/*.yaml and same experiment keys (only targets differ)."""
    import yaml
    mine = C.load_config()
    for part, fn in (("denoiser", "denoiser.yaml"), ("motion_vae", "motion_vae.yaml"), ("scheduler", "scheduler.yaml")):
        ref = yaml.safe_load(open(os.path.join(REF, "configs", "modules", fn)))[part]
        got = yaml.safe_load(open(os.path.join(C.CONFIG_DIR, "modules_hip", fn)))[part]
        assert ref["params"] == got["params"], part
        for k in ("num_inference_timesteps", "eta"):
            if k in ref:
                assert ref[k] == got[k]
    ref_exp = yaml.safe_load(open(os.path.join(REF, "configs", "config_mld_humanml3d.yaml")))
    for k in ("latent_dim", "ff_size", "num_layers", "num_head", "guidance_scale", "guidance_uncondp", "condition", "vae"):
        assert ref_exp["model"][k] == mine.model[k], k
    assert ref_exp["TRAIN"]["ABLATION"] == {k: mine.TRAIN.ABLATION[k] for k in ref_exp["TRAIN"]["ABLATION"]}
    assert ref_exp["TEST"]["CHECKPOINTS"] == mine.TEST.CHECKPOINTS


def test_instantiate_from_config_contract():
    with pytest.raises(KeyError):
        C.instantiate_from_config({"params": {}})
    s = C.instantiate_from_config(C.load_config().model.scheduler)
    assert isinstance(s, HipDDIMScheduler) and s.init_noise_sigma == 1.0


# ------------------------------------------------------------------------------------------ scheduler
def test_scheduler_matches_oracle_tables():
    s = HipDDIMScheduler(**C.load_config().model.scheduler.params)
    o = O.DDIMSchedule()
    s.set_timesteps(50)
    np.testing.assert_array_equal(s.timesteps.numpy(), o.set_timesteps(50))
    np.testing.assert_allclose(s.alphas_cumprod.numpy(), o.alphas_cumprod, rtol=3e-6)
    x, e = torch.randn(3, 1, 256), torch.randn(3, 1, 256)
    for t in (981, 501, 1):
        np.testing.assert_allclose(s.step(e, t, x, eta=0.0).prev_sample.numpy(), o.step(e.numpy(), t, x.numpy()), atol=2e-6)
    with pytest.raises(NotImplementedError):
        s.step(e, 1, x, eta=0.5)
    import inspect
    assert "eta" in inspect.signature(s.step).parameters            # the reference probes for it (mld.py:318-320)
    assert s.config.num_train_timesteps == 1000


# ------------------------------------------------------------------------------------------ modules
def test_state_dict_keys_match_the_reference_modules(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    cfg = C.load_config()
    den = C.instantiate_from_config(cfg.model.denoiser)
    vae = C.instantiate_from_config(cfg.model.motion_vae)
    assert {k: list(v.shape) for k, v in den.state_dict().items()} == keys["denoiser"]
    assert {k: list(v.shape) for k, v in vae.state_dict().items()} == keys["vae"]
    # strict load of a foreign checkpoint works and survives a round trip
    sd = {k: torch.randn(*s) for k, s in keys["denoiser"].items()}
    den.load_state_dict(sd, strict=True)
    assert torch.equal(den.state_dict()["encoder.norm.weight"], sd["encoder.norm.weight"])
    with pytest.raises(RuntimeError):
        den.load_state_dict({k: v for k, v in sd.items() if k != "encoder.norm.weight"}, strict=True)


def test_unsupported_configurations_fail_loudly():
    abl = dict(SKIP_CONNECT=True, VAE_TYPE="mld", DIFF_PE_TYPE="mld", PE_TYPE="mld", MLP_DIST=False)
    with pytest.raises(NotImplementedError):
        HipMldDenoiser(ablation=abl, condition="image", num_layers=9)
    with pytest.raises(NotImplementedError):
        HipMldDenoiser(ablation={**abl, "VAE_TYPE": "no"}, num_layers=9)          # raw motion needs arch trans_dec + d=512
    with pytest.raises(NotImplementedError):
        HipMldVae(ablation=abl, nfeats=263, arch="all_encoder")
    d = HipMldDenoiser(ablation=abl, num_layers=9)
    with pytest.raises(RuntimeError):                      # CPU tensors, no injected engine: no CPU path
        d(torch.zeros(2, 1, 256), 5, torch.zeros(2, 1, 768))


@pytest.fixture(scope="module")
def sim_key():
    eng = simlib._lib.Engine(lib=simlib.sim_library(), use_graph=0, max_batch=4, max_frames=40, num_inference_steps=2, num_layers=simlib.SIM_LAYERS)
    key = E.inject_engine(eng, "inject:hostmirror")
    yield key
    E._engines.pop(key, None)
    eng.close()


def _cfg2():
    return C.load_config(overrides={"model.scheduler.num_inference_timesteps": 2, **TEXT_OVERRIDES})


def test_modules_forward_through_the_engine(sim_key):
    cfg = _cfg2()
    den = C.instantiate_from_config(cfg.model.denoiser).use_engine(sim_key)
    vae = C.instantiate_from_config(cfg.model.motion_vae).use_engine(sim_key)
    ops = O.NumpyOps(np.float32)
    b = syn.make_batch(2, [20, 13])
    x = torch.from_numpy(np.concatenate([b.init_latents] * 2))
    out = den(sample=x, timestep=torch.tensor(981), encoder_hidden_states=torch.from_numpy(b.text_emb), lengths=[20, 13] * 2)
    assert isinstance(out, tuple) and out[0].shape == (4, 1, 256)
    ref = O.denoiser_forward(ops, O.to_backend(ops, simlib.text_weights()[0]), x.numpy(), 981, b.text_emb)
    assert np.abs(out[0].numpy() - ref).max() < 5e-5
    z = torch.randn(1, 2, 256)
    feats = vae.decode(z, [20, 13])
    fr = O.vae_decode(ops, O.to_backend(ops, simlib.text_weights()[1]), z.permute(1, 0, 2).numpy(), [20, 13])
    assert feats.shape == (2, 20, 263) and np.abs(feats.numpy() - fr).max() < 5e-5
    # weights are re-uploaded after load_state_dict
    sd = den.state_dict()
    sd["encoder.norm.bias"] = sd["encoder.norm.bias"] + 1.0
    den.load_state_dict(sd)
    out2 = den(sample=x, timestep=981, encoder_hidden_states=torch.from_numpy(b.text_emb))[0]
    assert np.abs((out2 - out[0]).numpy() - 1.0).max() < 1e-5
    # encode (scope row 8f.1): (latent, Normal) like the reference, eps injectable
    fe = torch.randn(2, 20, 263)
    fe[1, 13:] = 0
    eps = torch.randn(2, 256)
    latent, dist = vae.encode(fe, [20, 13], eps=eps)
    lr, mr, lvr = O.vae_encode(ops, O.to_backend(ops, simlib.text_weights()[1]), fe.numpy(), [20, 13], eps.numpy()[:, None, :])
    assert latent.shape == (1, 2, 256) and isinstance(dist, torch.distributions.Normal)
    assert np.abs(dist.loc[0].numpy() - mr[:, 0]).max() < 5e-5
    assert np.abs(dist.scale[0].numpy() - np.sqrt(np.exp(lvr[:, 0]))).max() < 5e-5
    assert np.abs(latent[0].numpy() - lr[:, 0]).max() < 1e-4


def test_mld_forward_fused_and_modular_agree_with_oracle(sim_key):
    cfg = _cfg2()
    dm = HipDataModule(cfg, engine_key=sim_key)
    enc = SyntheticTextEncoder()
    model = MLD(cfg, dm, text_encoder=enc, engine_key=sim_key).eval()
    texts, lengths = ["a man kicks with his left leg.", "a person walks backward slowly."], [24, 17]
    lat0 = torch.from_numpy(syn.make_batch(2, lengths).init_latents)
    assert model.fused
    joints = model({"text": texts, "length": lengths}, init_latents=lat0)
    assert [tuple(j.shape) for j in joints] == [(24, 22, 3), (17, 22, 3)]
    # oracle on the same embeddings / noise
    ops = O.NumpyOps(np.float32)
    emb = enc([""] * 2 + texts).numpy()
    assert (emb[0] == emb[1]).all()
    mean, std = syn.make_mean_std()
    jr = O.sample(ops, O.to_backend(ops, simlib.text_weights()[0]), O.to_backend(ops, simlib.text_weights()[1]),
                  emb, lat0.numpy(), lengths, mean, std, steps=2)
    for i, n in enumerate(lengths):
        assert np.abs(joints[i].numpy() - jr[i, :n]).max() < 1e-4
    # the reference-style Python loop over the drop-in parts gives the same motions
    z = model._diffusion_reverse(torch.from_numpy(emb), lengths, init_latents=lat0)
    assert z.shape == (1, 2, 256)
    feats = model.vae.decode(z.contiguous(), lengths)
    j2 = model.feats2joints(feats)
    assert np.abs(j2.numpy() - jr).max() < 1e-4
    # condition 'text_uncond' (mld.py:228-229): the same network, empty prompts on both CFG halves -> text-independent motions
    cfg_u = C.load_config(overrides={"model.scheduler.num_inference_timesteps": 2, "model.condition": "text_uncond", **TEXT_OVERRIDES})
    mu = MLD(cfg_u, dm, text_encoder=enc, engine_key=sim_key).eval()
    ja = mu({"text": texts, "length": lengths}, init_latents=lat0)
    jb = mu({"text": ["something else entirely", "x"], "length": lengths}, init_latents=lat0)
    assert all(torch.equal(a, b) for a, b in zip(ja, jb))
    # recon_from_motion (mld.py:277-288): encode -> decode -> joints, plus the joints of the reference motion
    fr = torch.randn(2, 24, 263) * 0.3
    fr[1, 17:] = 0
    torch.manual_seed(3)
    jr_rec, jr_ref = model.recon_from_motion({"motion": fr, "length": lengths})
    assert [tuple(j.shape) for j in jr_rec] == [(24, 22, 3), (17, 22, 3)] and [tuple(j.shape) for j in jr_ref] == [(24, 22, 3), (17, 22, 3)]
    assert np.abs(jr_ref[1].numpy() - O.feats2joints(ops, fr.numpy(), mean, std)[1, :17]).max() < 1e-4
    # checkpoint contract: denoiser.* / vae.* keys, text_encoder.* re-injected, t2m_* ignored (base.py:117-127)
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith("text_encoder.")}
    sd["t2m_textencoder.fake"] = torch.zeros(1)
    model.load_state_dict(sd, strict=True)
    assert all(k.split(".")[0] in ("denoiser", "vae", "text_encoder") for k in model.state_dict())


# ------------------------------------------------------------------------------------------ action variant (config 5)
A2M_CFG = os.path.join(C.CONFIG_DIR, "config_mld_humanact12.yaml")


def test_action_config_and_reference_yaml_parity():
    cfg = C.load_config(A2M_CFG)
    assert cfg.model.condition == "action" and cfg.model.denoiser.target == "mld_hip.denoiser.HipMldDenoiser"
    assert cfg.model.denoiser.params.num_layers == 15 and cfg.model.denoiser.params.nclasses == 12
    assert cfg.model.motion_vae.target == "mld_hip.vae.HipActorVae" and cfg.model.motion_vae.params.num_layers == 6
    assert cfg.model.motion_vae.params.nfeats == 150
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present")
    import yaml
    for part, fn in (("denoiser", "denoiser.yaml"), ("motion_vae", "motion_vae.yaml"), ("scheduler", "scheduler.yaml")):
        ref = yaml.safe_load(open(os.path.join(REF, "configs", "modules_humanact12", fn)))[part]
        got = yaml.safe_load(open(os.path.join(C.CONFIG_DIR, "modules_hip_humanact12", fn)))[part]
        assert ref["params"] == got["params"], part
    ref_exp = yaml.safe_load(open(os.path.join(REF, "configs", "config_mld_humanact12.yaml")))
    for k in ("latent_dim", "ff_size", "num_layers", "num_head", "guidance_scale", "guidance_uncondp", "condition", "vae"):
        assert ref_exp["model"][k] == cfg.model[k], k
    assert ref_exp["TRAIN"]["ABLATION"] == {k: cfg.TRAIN.ABLATION[k] for k in ref_exp["TRAIN"]["ABLATION"]}


def test_action_state_dict_keys_match_the_reference_modules(golden_dir):
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    cfg = C.load_config(A2M_CFG)
    den = C.instantiate_from_config(cfg.model.denoiser)
    vae = C.instantiate_from_config(cfg.model.motion_vae)
    assert isinstance(vae, HipActorVae)
    assert {k: list(v.shape) for k, v in den.state_dict().items()} == keys["denoiser_action"]
    assert {k: list(v.shape) for k, v in vae.state_dict().items()} == keys["actor_vae"]


def test_action_mld_fused_and_modular_agree_with_oracle():
    eng = simlib.sim_action_engine(max_batch=4, max_frames=24, num_inference_steps=2)
    key = E.inject_engine(eng, "inject:hostmirror_a2m")
    try:
        cfg = C.load_config(A2M_CFG, overrides={"model.scheduler.num_inference_timesteps": 2, **simlib.ACTION_OVERRIDES})
        dm = HipDataModule(cfg, nfeats=150, njoints=25, name="humanact12", engine_key=key)
        model = MLD(cfg, dm, engine_key=key).eval()
        assert model.fused and model.text_encoder is None and model.vae_type == "actor"
        sdd, sdv = simlib.action_weights()
        model.denoiser.load_state_dict({k: torch.from_numpy(v) for k, v in sdd.items()}, strict=True)
        model.vae.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()}, strict=True)
        acts, lat0, _ = syn.make_action_batch(3, 16)
        lengths = [16, 9, 16]
        batch = {"action": torch.from_numpy(acts.astype(np.int64))[:, None], "length": lengths}
        rs = model.a2m_eval(batch, init_latents=torch.from_numpy(lat0))
        assert rs["m_rst"].shape == (3, 16, 150) and rs["m_lens"] == lengths
        ops = O.NumpyOps(np.float32)
        fr = O.sample_action(ops, O.to_backend(ops, sdd), O.to_backend(ops, sdv), acts, lat0, lengths, steps=2)
        assert np.abs(rs["m_rst"].numpy() - fr).max() < 1e-4
        # the reference-style loop over the per-op drop-ins (cond = cat(zeros, actions), mld.py:716-717)
        a = batch["action"]
        z = model._diffusion_reverse(torch.cat((torch.zeros_like(a), a)), lengths, init_latents=torch.from_numpy(lat0))
        f2 = model.vae.decode(z.contiguous(), lengths)
        assert np.abs(f2.numpy() - fr).max() < 1e-4
        with pytest.raises(NotImplementedError):
            model.feats2joints(f2)                                  # SMPL layout: out of scope, fails loudly
        # ActorVae.encode through the drop-in: (latent [1,B,D], Normal(mu, std)) like the reference
        fe = torch.randn(2, 16, 150)
        fe[1, 9:] = 0
        eps = torch.randn(2, 256)
        latent, dist = model.vae.encode(fe, [16, 9], eps=eps)
        lr, mr, lvr = O.actor_encode(ops, O.to_backend(ops, sdv), fe.numpy(), [16, 9], eps.numpy()[:, None, :])
        assert latent.shape == (1, 2, 256) and isinstance(dist, torch.distributions.Normal) and dist.loc.shape == (2, 256)
        assert np.abs(dist.loc.numpy() - mr[:, 0]).max() < 5e-5
        assert np.abs(dist.scale.numpy() - np.sqrt(np.exp(lvr[:, 0]))).max() < 5e-5
        assert np.abs(latent[0].numpy() - lr[:, 0]).max() < 1e-4
        # a text-variant module must refuse an action engine instead of mis-loading
        with pytest.raises(RuntimeError):
            C.instantiate_from_config(C.load_config().model.denoiser).use_engine(key)(
                torch.zeros(2, 1, 256), 5, torch.zeros(2, 1, 768))
    finally:
        E._engines.pop(key, None)
        eng.close()


# ------------------------------------------------------------------------------------------ diffusion-only variant (config 4)
NOVAE_CFG = os.path.join(C.CONFIG_DIR, "config_novae_humanml3d.yaml")


def test_novae_config_and_reference_yaml_parity(golden_dir):
    cfg = C.load_config(NOVAE_CFG)
    assert cfg.model.vae_type == "no" and cfg.model.denoiser.params.arch == "trans_dec"
    assert cfg.model.denoiser.params.ablation.VAE_TYPE == "no" and cfg.model.latent_dim == [1, 512]
    assert cfg.model.scheduler.target == "mld_hip.scheduler.HipDDPMScheduler" and cfg.model.scheduler.num_inference_timesteps == 1000
    den = C.instantiate_from_config(cfg.model.denoiser)
    keys = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    assert {k: list(v.shape) for k, v in den.state_dict().items()} == keys["denoiser_novae"]
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present")
    import yaml
    for part, fn in (("denoiser", "denoiser.yaml"), ("scheduler", "scheduler.yaml")):
        ref = yaml.safe_load(open(os.path.join(REF, "configs", "modules_novae", fn)))[part]
        got = yaml.safe_load(open(os.path.join(C.CONFIG_DIR, "modules_hip_novae", fn)))[part]
        assert ref["params"] == got["params"], part
        if part == "scheduler":
            assert ref["num_inference_timesteps"] == got["num_inference_timesteps"]
    ref_exp = yaml.safe_load(open(os.path.join(REF, "configs", "config_novae_humanml3d.yaml")))
    for k in ("latent_dim", "ff_size", "num_layers", "num_head", "guidance_scale", "guidance_uncondp", "condition", "vae", "vae_type"):
        assert ref_exp["model"][k] == cfg.model[k], k


def test_ddpm_scheduler_matches_oracle_tables(golden_dir):
    s = HipDDPMScheduler(**C.load_config(NOVAE_CFG).model.scheduler.params)
    o = O.DDPMSchedule()
    s.set_timesteps(1000)
    np.testing.assert_array_equal(s.timesteps.numpy(), o.set_timesteps(1000))
    tab = np.load(os.path.join(golden_dir, "ddpm_table.npz"))["coeffs"]
    for t in (999, 500, 1, 0):
        np.testing.assert_allclose(np.array(s.coeffs(t), np.float32), tab[t], rtol=1e-6)
        np.testing.assert_allclose(np.array(o.coeffs(t), np.float32), tab[t], rtol=0)
    x, e, nz = torch.randn(2, 5, 263), torch.randn(2, 5, 263), torch.randn(2, 5, 263)
    np.testing.assert_allclose(s.step(e, 500, x, noise=nz).prev_sample.numpy(), o.step(e.numpy(), 500, x.numpy(), nz.numpy()), atol=2e-6)
    np.testing.assert_allclose(s.step(e, 0, x).prev_sample.numpy(), o.step(e.numpy(), 0, x.numpy()), atol=2e-6)   # no noise at t = 0
    import inspect
    assert "eta" not in inspect.signature(s.step).parameters          # the reference probes for it (mld.py:318-320)


def test_novae_mld_fused_and_modular_agree_with_oracle():
    eng = simlib.sim_novae_engine(num_layers=2, max_batch=2, max_frames=24, num_inference_steps=2)
    key = E.inject_engine(eng, "inject:hostmirror_novae")
    try:
        cfg = C.load_config(NOVAE_CFG, overrides={"model.scheduler.num_inference_timesteps": 2, "model.denoiser.params.num_layers": 2})
        dm = HipDataModule(cfg, engine_key=key)
        enc = SyntheticTextEncoder()
        model = MLD(cfg, dm, text_encoder=enc, engine_key=key).eval()
        assert model.vae is None and model.vae_type == "no" and model.fused
        texts, lengths = ["a man kicks with his left leg.", "a person walks backward slowly."], [12, 7]
        g = syn._rng(21, "hm_novae")
        lat0 = torch.from_numpy(g.standard_normal((2, 12, 263)).astype(np.float32))
        noise = torch.from_numpy(g.standard_normal((2, 2, 12, 263)).astype(np.float32))
        joints = model({"text": texts, "length": lengths}, init_latents=lat0, step_noise=noise)
        assert [tuple(j.shape) for j in joints] == [(12, 22, 3), (7, 22, 3)]
        ops = O.NumpyOps(np.float32)
        sd = syn.make_novae_denoiser_state_dict(dims=syn.ModelDims(latent_dim=512, num_layers=2))
        emb = enc([""] * 2 + texts).numpy()
        mean, std = syn.make_mean_std()
        jr, fr = O.sample_novae(ops, O.to_backend(ops, sd), emb, lat0.numpy(), lengths, noise.numpy(), mean, std, steps=2)
        for i, n in enumerate(lengths):
            assert np.abs(joints[i].numpy() - jr[i, :n]).max() < 2e-4
        # the reference-style loop over the per-op drop-ins (denoiser + HipDDPMScheduler.step with the same noise)
        z = model._diffusion_reverse(torch.from_numpy(emb), lengths, init_latents=lat0, step_noise=noise)
        assert z.shape == (12, 2, 263)
        assert np.abs(z.permute(1, 0, 2).numpy() - fr).max() < 2e-4
        # without injected noise the engine's Philox stream is used: reproducible per seed, different across seeds
        j1, f1 = model.sample_novae(torch.from_numpy(emb), lengths, lat0, seed=5)
        j2, f2 = model.sample_novae(torch.from_numpy(emb), lengths, lat0, seed=5)
        j3, f3 = model.sample_novae(torch.from_numpy(emb), lengths, lat0, seed=6)
        assert torch.equal(f1, f2) and (f1 - f3).abs().max() > 1e-3
    finally:
        E._engines.pop(key, None)
        eng.close()


def test_demo_example_parser(tmp_path):
    from mld_hip.demo import load_example_input
    p = tmp_path / "ex.txt"
    p.write_text("50 a man kicks with something or someone with his left leg.\n100 A person is skipping rope.\n")
    texts, lens = load_example_input(str(p))
    assert lens == [50, 100] and texts[1] == "A person is skipping rope."


def test_demo_writer_file_contract(tmp_path, sim_key):
    """The demo's output contract (reference demo.py:166-194): for a "<length> <prompt>" example file, one
    ``Example_<length>_batch0_<i>.npy`` of shape (length_i, 22, 3) float32 per line plus the prompt as ``.txt`` beside it."""
    from mld_hip.demo import load_example_input, write_motions
    ex = tmp_path / "example.txt"
    ex.write_text("20 a man kicks with his left leg.\n13 A person is skipping rope.\n\n")
    texts, lengths = load_example_input(str(ex))
    cfg = _cfg2()
    model = MLD(cfg, HipDataModule(cfg, engine_key=sim_key), text_encoder=SyntheticTextEncoder(), engine_key=sim_key).eval()
    sdd, sdv = simlib.text_weights()
    model.denoiser.load_state_dict({k: torch.from_numpy(v) for k, v in sdd.items()}, strict=True)
    model.vae.load_state_dict({k: torch.from_numpy(v) for k, v in sdv.items()}, strict=True)
    out = tmp_path / "results"
    paths = write_motions(model, texts, lengths, str(out), replication=2, log=lambda *a: None)
    names = sorted(os.listdir(out))
    assert names == sorted(f"Example_{n}_batch{r}_{i}.{ext}" for r in range(2) for i, n in enumerate(lengths) for ext in ("npy", "txt"))
    assert [os.path.basename(p) for p in paths[:2]] == ["Example_20_batch0_0.npy", "Example_13_batch0_1.npy"]      # the reference's names
    for i, n in enumerate(lengths):
        j = np.load(out / f"Example_{n}_batch0_{i}.npy")
        assert j.shape == (n, 22, 3) and j.dtype == np.float32 and np.isfinite(j).all()
        assert (out / f"Example_{n}_batch0_{i}.txt").read_text() == texts[i]
    assert not np.array_equal(np.load(out / "Example_20_batch0_0.npy"), np.load(out / "Example_20_batch1_0.npy"))   # a fresh draw per replication


def test_lightning_checkpoint_is_readable_without_lightning(tmp_path):
    """A checkpoint whose pickle references classes of packages that are not installed (as the released Lightning files do
    with omegaconf / pytorch_lightning) still yields its state_dict; plain torch.load cannot read it."""
    import sys
    import types
    from mld_hip.checkpoint import load_lightning_state_dict

    mod = types.ModuleType("fake_lightning_pkg")

    class DictConfig(dict):
        pass

    class ModelCheckpoint:
        def __init__(self):
            self.best = 0.41
    DictConfig.__module__ = ModelCheckpoint.__module__ = "fake_lightning_pkg"
    DictConfig.__qualname__, ModelCheckpoint.__qualname__ = "DictConfig", "ModelCheckpoint"
    mod.DictConfig, mod.ModelCheckpoint = DictConfig, ModelCheckpoint
    sys.modules["fake_lightning_pkg"] = mod
    sd = {"denoiser.encoder.norm.weight": torch.arange(4.0), "vae.final_layer.bias": torch.ones(3), "t2m_moveencoder.w": torch.zeros(2)}
    path = str(tmp_path / "fake.ckpt")
    try:
        torch.save({"state_dict": sd, "hyper_parameters": DictConfig(a=1), "callbacks": {"ckpt": ModelCheckpoint()},
                    "pytorch-lightning_version": "1.7.7", "epoch": 5}, path)
    finally:
        del sys.modules["fake_lightning_pkg"]
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)            # ModuleNotFoundError: fake_lightning_pkg
    got, missing = load_lightning_state_dict(path, report_missing=True)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert missing == ["fake_lightning_pkg.DictConfig", "fake_lightning_pkg.ModelCheckpoint"]
    flat = str(tmp_path / "flat.pt")
    torch.save(sd, flat)
    assert set(load_lightning_state_dict(flat)) == set(sd)                   # a bare state dict is accepted too
